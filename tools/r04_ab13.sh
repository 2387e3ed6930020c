#!/bin/bash
# round 4: the leaf rule of k_extend_tlas_wide (an expensive test class waits for FW_LEAF_MIN lanes): part2 @256 and random_spheres
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -4; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
for i in 1 2; do
  for v in two leaf8 leaf12 leaf20; do FIREWORK_LIB=$V/lib_$v.so run "part2@256 $v" "--config C5_part2_all --spp 256" 3; done
done 2>&1 | tee $OUT/leaf_rule.txt
for v in two leaf12; do FIREWORK_LIB=$V/lib_$v.so run "C1 $v" "--config C1_random_spheres" 20; FIREWORK_LIB=$V/lib_$v.so run "volume@128 $v" "--config C4b_volume_test --spp 128" 3; done 2>&1 | tee -a $OUT/leaf_rule.txt
