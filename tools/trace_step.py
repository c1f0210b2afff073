#!/usr/bin/env python3
"""Development tool: the kernels of the last timed batch (from one K5r to the next) in a rocprofv3 --kernel-trace CSV:
start, duration and idle gap before each, in microseconds.   python tools/trace_step.py <kernel_trace.csv>"""
import csv
import sys

tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(tr) if 'rerank_kernel' in r['Kernel_Name']]
if len(ends) < 3:
    raise SystemExit("fewer than three batches in the trace")
# bench.py ends with its single-query loop and one more batch: take the last batch of the timed run
batch = tr[ends[-3] + 1:ends[-2] + 1]
t0 = int(tr[ends[-3]]['End_Timestamp'])
prev = t0
for r in batch:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:58]:58s} start {(st - t0) / 1e3:8.1f} dur {(en - st) / 1e3:7.1f} gap {(st - prev) / 1e3:6.1f}")
    prev = en
print(f"batch: {(prev - t0) / 1e3:.1f} us from the end of the previous batch's last kernel")
