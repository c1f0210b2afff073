import json,sys
j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["config"]["rows"], "ms/step", j["ms_per_step"], "main launch_ms", r["launch_ms"], "all_scan_ms", r["all_scan_ms"], "select_ms", r["select_ms"], "qps", j["value"])
