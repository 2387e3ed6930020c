#!/bin/bash
# round 4: cornell's second fetch: can it be made to hit in L2?  (non-temporal queue stores, earlier runs of the box lists, fewer resident waves, no lists at all)
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
for i in 1 2 3; do
  run "default   " "" 8
  for v in ntst run32 run48 defer6 ntst_run32; do FIREWORK_LIB=$V/lib_$v.so run "$v     " "" 8; done
  FIREWORK_NO_DEFER=1 run "no_defer  " "" 8
done 2>&1 | tee $OUT/cornell_refetch.txt
