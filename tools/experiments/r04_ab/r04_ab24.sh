#!/bin/bash
# round 4: hdri / volume — plain rectangles past the generic dispatch in the generic linear scans (no shared reciprocals)
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2 3; do
  for cfg in "hdri:--config C4a_hdri_test:4" "volume:--config C4b_volume_test:4"; do
    n=${cfg%%:*}; rest=${cfg#*:}; a=${rest%:*}; st=${rest##*:}
    FIREWORK_LIB=$V/lib_v1.so run "$n v1       " "$a" $st
    FIREWORK_LIB=$V/lib_lindirect.so run "$n lindirect" "$a" $st
  done
done 2>&1 | tee $OUT/lindirect.txt
