#!/bin/bash
# upper bound of what the rejection loop of random_in_unit_sphere costs k_shade: a variant capped at ONE attempt (wrong frames, timing only)
set -o pipefail
[ -f firework_amd/lib/variants/lib_reject1.so ] || bash tools/build_variant.sh reject1 -DFW_MAX_REJECT=1 -Iinclude
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
run() { timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>$OUT/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))" || tail -5 $OUT/err.txt; }
for cfg in "--config C3_suzanne" "--config C5_part2_all --spp 64" "" "--config C1_random_spheres"; do
  for i in 1 2; do run "loop   $cfg" "$cfg"; FIREWORK_LIB=$R/firework_amd/lib/variants/lib_reject1.so run "capped $cfg" "$cfg"; done
done 2>&1 | tee $OUT/reject_ab.txt
