#!/bin/bash
# interleaved timings of every library variant in firework_amd/lib/variants against lib_base:  tools/r03_variants.sh <tag> "<bench args>" ...
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
run() { timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2))"; }
for rep in 1 2 3; do for f in $R/firework_amd/lib/variants/lib_*.so; do v=$(basename $f .so); export FIREWORK_LIB=$f
  for a in "$@"; do run $v "$a"; done; done; done 2>&1 | tee $OUT/variants_ab.txt
