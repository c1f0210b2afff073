#!/bin/bash
# development: kernel timeline (one batch in flight) of a bench configuration; usage: tools/trace_kernels.sh <outdir> [bench args]
out=$1; shift
export TMPDIR=/tmp
VSR_BENCH_SESSIONS=1 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 bench.py --legs prefilter --steps 6 --warmup 2 --no-cpu-baseline --sustained-s 0 "$@" > /dev/null 2>&1
python3 tools/trace_summary.py "$out" | grep -v rocclr | tail -6
