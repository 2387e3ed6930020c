#!/bin/bash
# round 4: cornell, the box-list scan capped at 5 / 4 waves per SIMD by LDS (so that k_shade of the other batch finds registers), with and without stagger
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
for i in 1 2 3; do
  run "default              " "" 8
  FIREWORK_LIB=$V/lib_ab_pad3584.so run "5 waves              " "" 8
  FIREWORK_LIB=$V/lib_ab_pad3584.so FIREWORK_STAGGER=1 run "5 waves stagger      " "" 8
  FIREWORK_LIB=$V/lib_ab_pad5632.so run "4 waves              " "" 8
  FIREWORK_LIB=$V/lib_ab_pad5632.so FIREWORK_STAGGER=1 run "4 waves stagger      " "" 8
  FIREWORK_LIB=$V/lib_ab_pad5632.so FIREWORK_STAGGER=1 FIREWORK_PATHS_PER_BATCH=134217728 run "4 waves stagger 4 bat" "" 8
done 2>&1 | tee $OUT/cornell_caps.txt
