#!/bin/bash
# round 3: the exact walk — suite, default-mode divergence, EXACT_ALL divergence (the renderer as the reference's own traversal), timing with and without
set -o pipefail
OUT=gpurun_out/r03e; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for c in C3_suzanne:1280:720:16 C5_part2_all:1920:1080:4 C1_random_spheres:400:225:64; do
  IFS=: read -r NAME W H SPP <<< "$c"
  timeout -k 10 400 python tools/diverge.py $NAME $W $H $SPP --max-pixels 4 --tol 1e-4 --out $OUT/diverge_$NAME.json > $OUT/diverge_$NAME.log 2>&1; echo "$NAME rc=$?"
  tail -n 1 $OUT/diverge_$NAME.log | cut -c1-500
  FIREWORK_EXACT_ALL=1 timeout -k 10 600 python tools/diverge.py $NAME $W $H $SPP --max-pixels 2 --tol 1e-4 --out $OUT/diverge_all_$NAME.json > $OUT/diverge_all_$NAME.log 2>&1; echo "$NAME EXACT_ALL rc=$?"
  tail -n 1 $OUT/diverge_all_$NAME.log | cut -c1-500
done
# teapot (rotated meshes: the ill-direction test in the meshes' frames) at its example size, reduced spp
timeout -k 10 400 python tools/diverge.py teapot 1920 1080 4 --max-pixels 4 --tol 1e-4 --out $OUT/diverge_teapot.json > $OUT/diverge_teapot.log 2>&1; echo "teapot rc=$?"; tail -n 1 $OUT/diverge_teapot.log | cut -c1-500
for v in 0 1; do
  echo "== FIREWORK_NO_EXACT=$v"
  for cfg in "C3_suzanne 2 1" "C5_part2_all 1 0" "C1_random_spheres 3 1"; do
    set -- $cfg
    if [ $v = 1 ]; then export FIREWORK_NO_EXACT=1; else unset FIREWORK_NO_EXACT; fi
    timeout -k 10 300 python3 bench.py --config $1 --steps $2 --warmup $3 --no-cpu-baseline --no-one-shot 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print(json.dumps({'config': d['config']['workload'], 'ms_per_frame': round(d['ms_per_step'],2), 'ms_extend': round(k.get('ms_extend',0),2), 'ms_shade': round(k.get('ms_shade',0),2)}))"
  done
done 2>&1 | tee $OUT/timing.txt
