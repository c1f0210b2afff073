# development: the headline step under variant environments (quick flags), one line per variant
F="--steps 20 --warmup 5 --legs prefilter --wiki-rows 0 --no-bf16-line --no-cpu-baseline --ivf-rows 0 --sustained-s 1"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1],"ms",d["ms_per_step"],"value",d["value"],"frac",r["frac"],"alone",r["alone"]["launch_ms"],"sust",d["sustained"]["value"],"kernel",r["kernel"][:40],"pass_rows",r["pass_rows_per_launch"])'
python bench.py $F 2>/dev/null | python -c "$P" base
VSR_K2I=1 python bench.py $F 2>/dev/null | python -c "$P" k2i
VSR_K2I=1 VSR_K2I_WIDE=1 python bench.py $F 2>/dev/null | python -c "$P" k2i_wide
