#!/bin/bash
# cornell, two batches in flight: cap the waves per CU of the scan and of k_shade (LDS floor per single-wave workgroup) so that both kernels are resident on every SIMD
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
run() { timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot --no-parity 2>$OUT/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms', round(d['ms_per_step'],2))" || tail -5 $OUT/err.txt; }
# 160 KB / floor = workgroups (= waves) per CU: 8192 -> 20 (5 per SIMD), 10240 -> 16 (4), 13653 -> 12 (3), 20480 -> 8 (2)
for combo in "0 0" "8192 0" "10240 0" "0 13653" "0 20480" "8192 13653" "10240 13653" "10240 20480" "13653 13653" "0 0"; do
  set -- $combo
  FIREWORK_SCAN_LDS=$1 FIREWORK_SHADE_LDS=$2 run "scan_lds=$1 shade_lds=$2"
done 2>&1 | tee $OUT/coreside.txt
