#!/bin/bash
# round 4: cornell with the ray carried in the box lists: suite for the scan, then 7 / 6 / 5 waves per SIMD
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -4; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
for i in 1 2 3; do
  run "waves7   " "" 8
  FIREWORK_LIB=$V/lib_defer6.so run "waves6   " "" 8
  FIREWORK_LIB=$V/lib_defer5.so run "waves5   " "" 8
done 2>&1 | tee $OUT/cornell_carry.txt
