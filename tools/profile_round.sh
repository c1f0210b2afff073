#!/bin/bash
# Round profile of the bench command, run on the GPU box: tools/profile_round.sh <outdir under gpurun_out/>
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (default flags)       -> stats/
#   2. SQ / TCC counter passes per timed leg (tools/pmc_run.sh: one rocprofv3 run per counter set, --kernel-trace only)
#   3. traffic.json: HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, keyed by bench.py's workload tag
set -e
out=$1
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/stats.log" || { echo "stats pass failed"; tail -5 "$out/stats.log"; }
echo "stats done"
for leg in prefilter postfilter; do
  tools/pmc_run.sh "$out/pmc_$leg" --legs $leg
  python3 tools/pmc_traffic.py "$out/pmc_$leg/pass4" "$out/pmc_$leg/pass5" "$out/traffic.json" "10000000x128 k=100 q=1000 $leg gpus=1" > "$out/traffic_$leg.txt"
done
echo "profile done"
