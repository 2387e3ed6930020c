#!/bin/bash
# pending-leaf variants of the LDS walks against the base build: parity tests under the variant, then interleaved timings
#   usage: tools/r03_pend.sh <tag> <variant> "<bench args>" ["<bench args>" ...]
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; V=$2; shift 2
FIREWORK_LIB=$R/firework_amd/lib/variants/lib_$V.so timeout -k 10 500 python -m pytest tests/test_gpu_divergence.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
run() { timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2))"; }
for rep in 1 2 3; do for v in base $V; do export FIREWORK_LIB=$R/firework_amd/lib/variants/lib_$v.so
  for a in "$@"; do run $v "$a"; done; done; done 2>&1 | tee $OUT/pend_ab.txt
