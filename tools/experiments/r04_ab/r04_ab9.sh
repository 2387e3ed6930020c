#!/bin/bash
# round 4: walk variants of k_blas_wide (two steps per exit check, exit rules), the second fetch of k_extend_linear_defer
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
echo "== k_blas_wide variants (suzanne, teapot @256)"
for i in 1 2 3; do
  run "default " "--config C3_suzanne" 4
  for v in two walk32 walk41; do FIREWORK_LIB=$V/lib_$v.so run "$v     " "--config C3_suzanne" 4; done
done 2>&1 | tee $OUT/wide_variants.txt
for i in 1 2; do run "default " "--config teapot --spp 256" 3; for v in two walk41; do FIREWORK_LIB=$V/lib_$v.so run "$v     " "--config teapot --spp 256" 3; done; done 2>&1 | tee -a $OUT/wide_variants.txt
echo "== cornell: the listed rays' second fetch (norefetch renders wrong frames: timing only)"
for i in 1 2 3; do run "default  " "" 8; FIREWORK_LIB=$V/lib_norefetch.so run "norefetch" "" 8; done 2>&1 | tee $OUT/norefetch.txt
