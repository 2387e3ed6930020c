#!/bin/bash
# round 4: the linear scan without the mesh walk (volume) and without the medium too (hdri) against the one generic kernel
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -3; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2 3; do
  for cfg in C4a_hdri_test C4b_volume_test; do
    FIREWORK_LIB=$V/lib_base.so run "$cfg generic scan " "--config $cfg" 4
    run "$cfg trimmed scan " "--config $cfg" 4
  done
done 2>&1 | tee $OUT/linear_variants.txt
