#!/bin/bash
# round 4: cornell — the walls' shared reciprocals (v1) and the box pre-tests reusing them (v2) against the build before
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot --no-parity $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in 1 2 3; do
  FIREWORK_LIB=$V/lib_base.so run "cornell before" "" 8
  FIREWORK_LIB=$V/lib_v1.so run "cornell v1    " "" 8
  FIREWORK_LIB=$V/lib_v2.so run "cornell v2    " "" 8
done 2>&1 | tee $OUT/cornell_rcp2.txt
FIREWORK_LIB=$V/lib_v2.so timeout -k 10 300 python3 tools/fuzz_defer.py 7000 200 2>&1 | tail -2
