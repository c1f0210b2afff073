#!/bin/bash
# development sweep: the headline leg of bench.py under different planner / kernel knobs (one process per configuration)
# usage: tools/sweep_env.sh <outdir> "ENV1=a ENV2=b" "ENV1=c" ...
out=$1; shift
mkdir -p "$out"
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg python3 bench.py --legs prefilter --steps 10 --warmup 3 --no-cpu-baseline --sustained-s 0 > "$out/cfg$i.json" 2> "$out/cfg$i.err"
  python3 - "$out/cfg$i.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:60s} ms/step {d['ms_per_step']:7.4f}  launch {r['launch_ms']:7.4f}  alone {r.get('alone', {}).get('launch_ms', 0):7.4f}  enq {d['host_enqueue_ms_per_step']:6.3f}  {r['kernel'][-30:]}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
