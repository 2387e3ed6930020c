#!/bin/bash
# round 4: cornell, batches per frame (2 = the default: one per lane) and the AB build's stagger, on the final box-list kernel
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
for i in 1 2 3; do
  run "2 batches        " "" 8
  FIREWORK_PATHS_PER_BATCH=134217728 run "4 batches        " "" 8
  FIREWORK_PATHS_PER_BATCH=67108864 run "8 batches        " "" 8
  FIREWORK_LIB=$V/lib_ab.so FIREWORK_STAGGER=1 run "2 batches stagger" "" 8
  FIREWORK_LIB=$V/lib_ab.so FIREWORK_STAGGER=1 FIREWORK_PATHS_PER_BATCH=134217728 run "4 batches stagger" "" 8
  FIREWORK_STREAMS=3 FIREWORK_PATHS_PER_BATCH=201326592 run "3 lanes 3 batches" "" 8
done 2>&1 | tee $OUT/cornell_batches.txt
