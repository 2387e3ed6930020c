#!/bin/bash
# round 4: k_extend_scan without the mesh walk and the medium's code where the scene has no medium (suzanne, teapot), 6 and 7 waves per SIMD
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tee $OUT/pytest.log | tail -3; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2 3; do
  FIREWORK_LIB=$V/lib_base.so run "suzanne generic scan      " "--config C3_suzanne" 3
  run "suzanne plain scan        " "--config C3_suzanne" 3
  FIREWORK_LIB=$V/lib_scan7.so run "suzanne plain scan 7 waves" "--config C3_suzanne" 3
  FIREWORK_LIB=$V/lib_base.so run "teapot@128 generic scan      " "--config teapot --spp 128" 3
  run "teapot@128 plain scan        " "--config teapot --spp 128" 3
  FIREWORK_LIB=$V/lib_scan7.so run "teapot@128 plain scan 7 waves" "--config teapot --spp 128" 3
done 2>&1 | tee $OUT/scan_variants.txt
