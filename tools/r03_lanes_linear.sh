R=$PWD
run() { timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', 'ms', round(d['ms_per_step'],2))"; }
for rep in 1 2 3; do for cfg in "--config C4a_hdri_test" "--config C4b_volume_test"; do for s in 1 2; do FIREWORK_STREAMS=$s run "streams=$s" "$cfg"; done; done; done
