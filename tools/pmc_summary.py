#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes: python tools/pmc_summary.py <out.json> <pass_dir>...
Every pass directory holds one rocprofv3 run (counter_collection.csv); counters of all passes are merged per kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for kern, ctrs in agg.items():
        res[kern] = {c: {"avg": sum(v) / len(v), "dispatches": len(v)} for c, v in ctrs.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for kern, ctrs in sorted(res.items()):
        if "vsr::" not in kern:
            continue
        print(kern[:110])
        for c, v in sorted(ctrs.items()):
            print(f"    {c:32s} {v['avg']:18.1f}  x{v['dispatches']}")


if __name__ == "__main__":
    main()
