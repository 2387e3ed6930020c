R=$PWD
run() {
  python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'Mrays/s', round(d['value']), 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'rg', round(k.get('ms_raygen',0),2), 'acc', round(k.get('ms_accumulate',0),2))"
}
for rep in 1 2; do for f in $R/firework_amd/lib/variants/lib_*.so; do v=$(basename $f .so); FIREWORK_LIB=$f run $v; done; done
FIREWORK_LIB=$R/firework_amd/lib/variants/lib_pk.so timeout -k 10 300 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
