#!/bin/bash
# round 4: one reciprocal per axis for plain rectangles and Rect3d faces — cornell, hdri, volume, part2; the linear scans at 7 and 6 waves
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -3; echo "pytest rc=$?"
timeout -k 10 600 python3 tools/fuzz_defer.py 5000 300 2>&1 | tail -2
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2 3; do
  for cfg in "cornell::8" "hdri:--config C4a_hdri_test:4" "volume:--config C4b_volume_test:4" "part2@256:--config C5_part2_all --spp 256:3" "suzanne:--config C3_suzanne:3"; do
    n=${cfg%%:*}; rest=${cfg#*:}; a=${rest%:*}; st=${rest##*:}
    FIREWORK_LIB=$V/lib_base.so run "$n before       " "$a" $st
    run "$n shared rcp   " "$a" $st
    case $n in hdri|volume) FIREWORK_LIB=$V/lib_lin6.so run "$n shared rcp 6w" "$a" $st;; esac
  done
done 2>&1 | tee $OUT/shared_rcp.txt
