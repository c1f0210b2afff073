# development: the headline step under variant environments (quick flags), one line per variant
F="--steps 20 --warmup 5 --legs prefilter --wiki-rows 0 --no-bf16-line --no-cpu-baseline --ivf-rows 0 --sustained-s 1"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1],"ms",d["ms_per_step"],"value",d["value"],"frac",r["frac"],"alone",r["alone"]["launch_ms"],"sust",d["sustained"]["value"])'
python bench.py $F 2>/dev/null | python -c "$P" base
for b in 2048 4096 6144 8192; do VSR_BLOCK_BUDGET=$b python bench.py $F 2>/dev/null | python -c "$P" budget$b; done
