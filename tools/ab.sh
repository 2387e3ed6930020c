R=$PWD
run() {
  python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'Mrays/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'ext', round(k.get('ms_extend',0),1), 'shd', round(k.get('ms_shade',0),1), 'rg', round(k.get('ms_raygen',0),1), 'acc', round(k.get('ms_accumulate',0),1))"
}
run default
for P in 33554432 67108864 134217728 268435456; do for W in 32768 65536 131072; do FIREWORK_WAVES=$W FIREWORK_PATHS_PER_BATCH=$P run P${P}_W$W; done; done
