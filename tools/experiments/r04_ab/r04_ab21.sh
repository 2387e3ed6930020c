#!/bin/bash
# round 4: cornell's wall rectangles with one reciprocal per axis instead of one per rectangle (k_extend_linear_defer's in-line loop)
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -3; echo "pytest rc=$?"
timeout -k 10 600 python3 tools/fuzz_defer.py 3000 300 2>&1 | tail -2
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2 3 4; do
  FIREWORK_LIB=$V/lib_base.so run "cornell before    " "" 8
  run "cornell shared rcp" "" 8
done 2>&1 | tee $OUT/cornell_rcp.txt
