#!/bin/bash
# round 4: scene-specialised extend kernels (no mesh walk / no medium where the scene has none) against the generic ones, all six configs
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -3; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2 3; do
  for cfg in "C1:--config C1_random_spheres:20" "cornell::6" "suzanne:--config C3_suzanne:3" "hdri:--config C4a_hdri_test:4" "volume:--config C4b_volume_test:4" "part2@256:--config C5_part2_all --spp 256:3" "teapot@128:--config teapot --spp 128:3"; do
    n=${cfg%%:*}; rest=${cfg#*:}; a=${rest%:*}; st=${rest##*:}
    FIREWORK_LIB=$V/lib_base.so run "$n generic    " "$a" $st
    run "$n specialised" "$a" $st
  done
done 2>&1 | tee $OUT/specialised.txt
