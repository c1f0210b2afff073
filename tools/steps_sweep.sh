F="--steps 20 --warmup 5 --legs prefilter --wiki-rows 0 --no-bf16-line --no-cpu-baseline --ivf-rows 0 --sustained-s 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1],"ms",d["ms_per_step"],"value",d["value"],"per_step",d["roofline"]["per_step"].get("hbm_traffic_frac"), "alone", d["roofline"]["alone"]["launch_ms"])'
python bench.py $F 2>/dev/null | python -c "$P" base
for s in 8 24 32; do VSR_SAMPLE_STRIDE=$s python bench.py $F 2>/dev/null | python -c "$P" stride$s; done
VSR_BENCH_SESSIONS=4 python bench.py $F 2>/dev/null | python -c "$P" sessions4
VSR_BENCH_SESSIONS=2 python bench.py $F 2>/dev/null | python -c "$P" sessions2
