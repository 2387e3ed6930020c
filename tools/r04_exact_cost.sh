#!/bin/bash
# round 4: what the exact walk costs each tree config — interleaved, FIREWORK_NO_EXACT=1 against the default (profiles/r04z_exact_cost.txt)
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'rays/sample', round(d['rays_per_sample'],3))"; }
for i in 1 2 3; do
  for cfg in "C3_suzanne:--config C3_suzanne:3" "part2@256:--config C5_part2_all --spp 256:3" "teapot@128:--config teapot --spp 128:3" "C1:--config C1_random_spheres:20"; do
    n=${cfg%%:*}; rest=${cfg#*:}; a=${rest%:*}; st=${rest##*:}
    run "$n exact on " "$a" $st
    FIREWORK_NO_EXACT=1 run "$n exact off" "$a" $st
  done
done 2>&1 | tee $OUT/exact_cost.txt
