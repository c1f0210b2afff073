F="--steps 20 --warmup 5 --legs prefilter,postfilter --wiki-rows 0 --no-bf16-line --no-cpu-baseline --ivf-rows 0 --sustained-s 1"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1],"ms",d["ms_per_step"],"value",d["value"],"frac",r["frac"],"launch_ms",r["launch_ms"],"alone",r["alone"]["launch_ms"],"post",d["postfilter"]["value"],d["postfilter"]["roofline"]["frac"],"sust",d["sustained"]["value"], (d.get("parity_spot_check") or {}).get("ids_and_distances_identical"))'
for l in 0 1; do VSR_SCAN_LANE=$l python bench.py $F 2>/dev/null | python -c "$P" lane$l; done
for s in 2 4; do VSR_SCAN_LANE=1 VSR_BENCH_SESSIONS=$s python bench.py $F 2>/dev/null | python -c "$P" lane1_sess$s; done
