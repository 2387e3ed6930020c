#!/bin/bash
# round 4: what the rejection loop of random_in_unit_sphere costs k_shade now (timing builds with 1 / 2 attempts at most: WRONG frames)
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'rays/sample', round(d['rays_per_sample'],2))"; }
for i in 1 2; do
  for v in default reject2 reject1; do
    L=$V/lib_$v.so; [ $v = default ] && L=$R/firework_amd/lib/libfirework_hip.so
    FIREWORK_LIB=$L run "cornell $v" "" 6
    FIREWORK_LIB=$L run "part2@256 $v" "--config C5_part2_all --spp 256" 3
    FIREWORK_LIB=$L run "suzanne $v" "--config C3_suzanne" 3
  done
done 2>&1 | tee $OUT/reject_cap.txt
