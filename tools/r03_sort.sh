#!/bin/bash
# round 3: ray sorting in the LDS walks — correctness, interleaved A/B, kernel stats
set -o pipefail
OUT=$PWD/gpurun_out/r03f; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
echo "== suzanne @64 A/B"; bash tools/ab_env.sh FIREWORK_NO_SORT "--config C3_suzanne --spp 64" 3 2>&1 | tee $OUT/ab_suzanne64.txt
echo "== suzanne full A/B"; bash tools/ab_env.sh FIREWORK_NO_SORT "--config C3_suzanne" 2 2>&1 | tee $OUT/ab_suzanne.txt
echo "== exact off A/B (suzanne @64)"; bash tools/ab_env.sh FIREWORK_NO_EXACT "--config C3_suzanne --spp 64" 2 2>&1 | tee $OUT/ab_exact64.txt
export TMPDIR=/tmp; cd /tmp
for v in sort nosort; do
  if [ $v = nosort ]; then export FIREWORK_NO_SORT=1; else unset FIREWORK_NO_SORT; fi
  rm -rf $OUT/prof_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -- python3 $GRAFT_REPO_ROOT/bench.py --config C3_suzanne --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot > $OUT/prof_$v.log 2>&1
  f=$(find $OUT/prof_$v -name "*kernel_stats.csv" | head -1); echo "== $v kernel stats"; head -12 $f | cut -c1-160
done
