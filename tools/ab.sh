R=$PWD
run() {
  python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'Mrays/s', round(d['value']), 'ms', round(d['ms_per_step'],2), 'dev', round(k.get('ms_render',0),2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'rg', round(k.get('ms_raygen',0),2), 'acc', round(k.get('ms_accumulate',0),2))"
}
run full
run spp512 "--spp 512"
run spp256 "--spp 256"
run spp128 "--spp 128"
run spp128_notiming "--spp 128 --no-kernel-timing"
