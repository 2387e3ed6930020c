#!/bin/bash
# round 4: cornell with never-overflowing ray-carrying box lists (drain between chunks) against the re-fetching lists
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -4; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
for i in 1 2 3; do
  run "carry    " "" 8
  FIREWORK_LIB=$V/lib_two.so run "refetch  " "" 8
done 2>&1 | tee $OUT/cornell_carry2.txt
