#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest.log | tail -6
bash tools/r03_flags.sh 2>&1 | tee $OUT/flags.txt
run() { timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
echo "== exact walk on / off"
for cfg in "--config C3_suzanne --spp 64" "--config C3_suzanne" "--config C5_part2_all --spp 16" "--config C1_random_spheres" "--config teapot --spp 32"; do
  for i in 1 2; do run "exact   $cfg" "$cfg"; FIREWORK_NO_EXACT=1 run "noexact $cfg" "$cfg"; done
done 2>&1 | tee $OUT/exact_ab.txt
for c in C3_suzanne:1280:720:16 C5_part2_all:1920:1080:4 teapot:1920:1080:4; do
  IFS=: read -r NAME W H SPP <<< "$c"
  timeout -k 10 400 python tools/diverge.py $NAME $W $H $SPP --max-pixels 2 --tol 1e-4 --out $OUT/diverge_$NAME.json > $OUT/diverge_$NAME.log 2>&1; echo "$NAME rc=$?"
  tail -n 1 $OUT/diverge_$NAME.log | cut -c1-400
done
