#!/bin/bash
# round 5: interleaved A/B of one runtime option on the product library:  tools/r05_env_ab.sh <outdir> <OPTION> reps -- "<bench args>" ...
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; OPT=$2; REPS=${3:-2}; shift 3; [ "$1" = "--" ] && shift
R=$PWD
run() { env $1 timeout -k 10 400 python3 $R/bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('%-24s %-34s' % ('$1', '$2'), 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
for cfg in "$@"; do for i in $(seq $REPS); do run "FIREWORK_$OPT=0" "$cfg"; run "FIREWORK_$OPT=1" "$cfg"; done; done 2>&1 | tee $OUT/env_ab_$OPT.txt
