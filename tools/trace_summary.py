#!/usr/bin/env python3
"""Per-kernel durations and the timeline of the last full batch from a rocprofv3 --kernel-trace CSV directory."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:72s} n={len(v):5d} avg={sum(v)/len(v):9.1f} us  min={min(v):8.1f}  total={sum(v)/1000:8.2f} ms")
main = [i for i, r in enumerate(rows) if "wide_kernel" in r["Kernel_Name"] and "true" not in r["Kernel_Name"].split("(")[0].split(",")[2]]
if len(main) >= 6:
    a, b = main[len(main) // 2], main[len(main) // 2 + 1]        # two consecutive batches from the middle of the timed loop
if len(main) >= 6:
    seg = rows[a - 3:b + 4]
    t0 = int(seg[0]["Start_Timestamp"])
    prev_end = t0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1000:9.1f} +{(e - s) / 1000:8.1f} gap {(s - prev_end) / 1000:7.1f}  q={r['Queue_Id']} {r['Kernel_Name'][:60]} grid={r['Grid_Size_X']}")
        prev_end = max(prev_end, e)
