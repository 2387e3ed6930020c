#!/bin/bash
# lanes (batches in flight) A/B on the tree-walk configs: FIREWORK_STREAMS=1..4
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
run() { timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>$OUT/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms', round(d['ms_per_step'],2))" || tail -5 $OUT/err.txt; }
for cfg in "--config C3_suzanne" "--config C5_part2_all --spp 256" "--config teapot --spp 256"; do
  for s in 1 2 3 4 2; do FIREWORK_STREAMS=$s run "streams=$s $cfg" "$cfg"; done
done 2>&1 | tee $OUT/lanes_ab.txt
