R=$PWD
run() {
  timeout -k 10 120 python3 $R/bench.py --config $2 --steps 2 --warmup 1 --no-cpu-baseline $3 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1 $2', 'Mrays/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'ext', round(k.get('ms_extend',0),1), 'shd', round(k.get('ms_shade',0),1))"
}
for f in $R/firework_amd/lib/variants/lib_*.so; do v=$(basename $f .so); FIREWORK_LIB=$f run $v C3_suzanne; FIREWORK_LIB=$f run $v C5_part2_all "--spp 128"; FIREWORK_LIB=$f run $v C1_random_spheres; done
FIREWORK_LIB=$R/firework_amd/lib/variants/lib_new.so timeout -k 10 300 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
