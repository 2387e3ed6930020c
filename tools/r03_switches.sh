#!/bin/bash
# The GPU parity tests under the runtime switches that select the other kernels (the exact walk and the NaN-ray rule must hold in all of them)
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
for sw in "FIREWORK_STREAMS=1" "FIREWORK_NO_LDS_TREES=1" "FIREWORK_TLAS_REFILL=0" "FIREWORK_BVH=median" "FIREWORK_NO_LDS_TRIS=1" "FIREWORK_STREAMS=4" "FIREWORK_SHADE_LIST=1" "FIREWORK_NO_HOIST=1"; do
  echo "== $sw"
  env $sw timeout -k 10 500 python -m pytest tests/test_gpu_divergence.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 || { echo "FAILED under $sw"; exit 1; }
done 2>&1 | tee $OUT/switches.txt
