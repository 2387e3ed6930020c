#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
run() { timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>$OUT/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))" || tail -5 $OUT/err.txt; }
echo "== exact walk on / off"
for cfg in "--config C3_suzanne --spp 64" "--config C3_suzanne" "--config C5_part2_all --spp 16" "--config C1_random_spheres" "--config teapot --spp 32"; do
  for i in 1 2; do run "exact   $cfg" "$cfg"; FIREWORK_NO_EXACT=1 run "noexact $cfg" "$cfg"; done
done 2>&1 | tee $OUT/exact_ab.txt
timeout -k 10 300 python3 tools/fuzz_many.py 1000 300 2>&1 | tail -3 | tee $OUT/fuzz.txt
