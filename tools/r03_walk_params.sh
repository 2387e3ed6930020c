#!/bin/bash
# the walk parameters of the LDS walks (refill thresholds, node-loop exit rule) re-swept after round 3's changes: variants in firework_amd/lib/variants
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
run() { timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2))"; }
for rep in 1 2; do for f in $R/firework_amd/lib/variants/lib_*.so; do v=$(basename $f .so); export FIREWORK_LIB=$f
  case $v in lib_t*) run $v "--config C5_part2_all --spp 64"; run $v "--config C1_random_spheres";; lib_b*) run $v "--config C3_suzanne"; run $v "--config C3_suzanne --spp 64";; esac
  if [ $v = lib_base ]; then run $v "--config C5_part2_all --spp 64"; run $v "--config C1_random_spheres"; fi
done; done 2>&1 | tee $OUT/walk_params.txt
