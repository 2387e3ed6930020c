#!/bin/bash
# round 4: idle lanes that trigger a refill in the wide walks (16): 4 / 8 / 32
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2; do
  for cfg in "suzanne:--config C3_suzanne:3" "teapot@128:--config teapot --spp 128:3" "part2@256:--config C5_part2_all --spp 256:3"; do
    n=${cfg%%:*}; rest=${cfg#*:}; a=${rest%:*}; st=${rest##*:}
    for v in base refill4 refill8 refill32; do FIREWORK_LIB=$V/lib_$v.so run "$n $v" "$a" $st; done
  done
done 2>&1 | tee $OUT/refill_min.txt
