#!/bin/bash
# round 3: GPU suite, divergence hunts, bench — tools/r03_step.sh <tag> [configs...]   (config = NAME:W:H:SPP)
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for c in "$@"; do
  IFS=: read -r NAME W H SPP <<< "$c"
  timeout -k 10 500 python tools/diverge.py $NAME $W $H $SPP --max-pixels 6 --tol 1e-4 --out $OUT/diverge_$NAME.json > $OUT/diverge_$NAME.log 2>&1; echo "$NAME rc=$?"
  tail -n 1 $OUT/diverge_$NAME.log | cut -c1-600
done
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-900 $OUT/bench.json
