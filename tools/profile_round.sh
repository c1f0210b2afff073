#!/bin/bash
# Round profile of the bench command, run on the GPU box in TWO gpurun calls (a call is limited to 20 minutes):
#   tools/profile_round.sh stats <outdir>   rocprofv3 --kernel-trace --stats of the driver's command (python3 bench.py --gpus 1 --steps 20
#                                           --warmup 5): per-kernel durations of every leg (SIFT10M, 768-d, hnsw, ivf, builds)
#   tools/profile_round.sh pmc <outdir>     SQ / TCC counter passes of the headline leg (tools/pmc_run.sh: one rocprofv3 run per counter
#                                           set, --kernel-trace only) + traffic.json (HBM bytes per launch, FETCH_SIZE / WRITE_SIZE passes),
#                                           then the same passes for the 768-d GEMM kernel through tools/kbench.py
set -e
what=$1
out=$2
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$what" = stats ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_under_rocprof.json" 2> "$out/stats.log" || { echo "stats pass failed"; tail -5 "$out/stats.log"; }
  find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/bench_kernel_stats.csv" \;
  find "$out/stats" -name "*.csv" -size +3M -delete
  echo "stats done"
else
  tools/pmc_run.sh "$out/pmc_prefilter" --legs prefilter --wiki-rows 0 --no-bf16-line
  python3 tools/pmc_traffic.py "$out/pmc_prefilter/pass4" "$out/pmc_prefilter/pass5" "$out/traffic.json" "10000000x128 k=100 q=1000 prefilter gpus=1" > "$out/traffic_prefilter.txt"
  tools/pmc_kbench.sh "$out/pmc_768" --rows 2000000 --dim 768 --gauss --metric cosine --cases full1000
  find "$out" -name "*.csv" -size +3M -delete
  echo "pmc done"
fi
