#!/usr/bin/env python3
"""Development tool: print the kernels of one steady-state step from a rocprofv3 --kernel-trace CSV (start, duration, gap)."""
import csv
import sys

tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 13
last = tr[-2 * n:-n]
t0 = int(last[0]['Start_Timestamp'])
prev = None
for r in last:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev) / 1e3 if prev else 0
    print(f"{r['Kernel_Name'][:58]:58s} start {(st - t0) / 1e3:8.1f} dur {(en - st) / 1e3:7.1f} gap {gap:6.1f}")
    prev = en
