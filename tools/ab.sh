R=$PWD
run() {
  python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'Mrays/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'ext', round(k.get('ms_extend',0),1), 'shd', round(k.get('ms_shade',0),1), 'rg', round(k.get('ms_raygen',0),1), 'acc', round(k.get('ms_accumulate',0),1))"
}
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
for S in 1 2 3 4; do FIREWORK_STREAMS=$S run streams$S; FIREWORK_STREAMS=$S run streams${S}_notiming --no-kernel-timing; done
