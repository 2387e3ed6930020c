#!/bin/bash
# round 4: k_shade without texture code where the chain state applies (hdri, volume: 38 -> 17 KB, no spills), five interleaved pairs
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -3; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2 3 4 5; do
  for cfg in "hdri:--config C4a_hdri_test:4" "volume:--config C4b_volume_test:4"; do
    n=${cfg%%:*}; rest=${cfg#*:}; a=${rest%:*}; st=${rest##*:}
    FIREWORK_LIB=$V/lib_base.so run "$n before" "$a" $st
    run "$n now   " "$a" $st
  done
done 2>&1 | tee $OUT/shade_notex.txt
