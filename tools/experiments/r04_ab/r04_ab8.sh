#!/bin/bash
# round 4: what the gating rule costs (noverify variant), triangles out of LDS, random scenes against the oracle
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
echo "== suzanne: default / tris from L2 / no gate rule (timing only) / both"
for i in 1 2 3; do
  run "default        " "--config C3_suzanne" 4
  FIREWORK_NO_LDS_TRIS=1 run "no_lds_tris    " "--config C3_suzanne" 4
  FIREWORK_LIB=$R/firework_amd/lib/variants/lib_noverify.so run "noverify       " "--config C3_suzanne" 4
  FIREWORK_NO_LDS_TRIS=1 FIREWORK_LIB=$R/firework_amd/lib/variants/lib_noverify.so run "noverify+l2tris" "--config C3_suzanne" 4
done 2>&1 | tee $OUT/suzanne_ab.txt
for i in 1 2; do
  run "default        " "--config teapot --spp 256" 3
  FIREWORK_LIB=$R/firework_amd/lib/variants/lib_noverify.so run "noverify       " "--config teapot --spp 256" 3
done 2>&1 | tee $OUT/teapot_ab.txt
echo "== random scenes"
timeout -k 10 500 python tools/fuzz_many.py 1000 600 2>&1 | tail -6 | tee $OUT/fuzz.txt | cut -c1-300
