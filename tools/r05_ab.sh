#!/bin/bash
# round 5: interleaved A/B of library variants (firework_amd/lib/variants/lib_<v>.so; "base" = the product library) on the configs given.
#   tools/r05_ab.sh <outdir> "<variants>" "<test variant or ->" [reps] -- "<bench args>" ...
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; VARS=$2; TESTV=$3; REPS=${4:-2}; shift 4; [ "$1" = "--" ] && shift
R=$PWD; V=$R/firework_amd/lib/variants
lib() { if [ "$1" = base ]; then echo $R/firework_amd/lib/libfirework_hip.so; else echo $V/lib_$1.so; fi; }
if [ "$TESTV" != "-" ]; then
  for t in $TESTV; do FIREWORK_LIB=$(lib $t) timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest_$t.log | tail -4; echo "pytest[$t] rc=$?"; done
fi
run() { FIREWORK_LIB=$(lib $1) timeout -k 10 400 python3 $R/bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('%-10s %-34s' % ('$1', '$2'), 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'rg', round(k.get('ms_raygen',0),2), 'Mrays/s', round(d['value']))"; }
for cfg in "$@"; do
  for i in $(seq $REPS); do for v in $VARS; do run $v "$cfg"; done; done
done 2>&1 | tee $OUT/ab.txt
