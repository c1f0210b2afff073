#!/usr/bin/env python3
"""Development tool: one-line digest of a bench.py JSON line read from stdin."""
import json
import sys

j = json.loads(sys.stdin.read())
r = j["roofline"]
print(j["config"]["rows"], "rows | ms/step", j["ms_per_step"], "| main launch ms", r["launch_ms"], "GB/s", r["achieved"],
      "| alone", (r.get("alone") or {}).get("launch_ms"), (r.get("alone") or {}).get("achieved"),
      "| host enqueue ms/step", j.get("host_enqueue_ms_per_step"), "| qps", j["value"])
