#!/bin/bash
# round 4: exact gating with tight, relaxed boxes: suite, wide vs pair, k_extend_scan at 5 waves, full-spp parity of the tree configs
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tee $OUT/pytest.log | tail -12; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
echo "== wide / pair / scan5"
for cfg in "--config C3_suzanne" "--config C5_part2_all --spp 256" "--config C1_random_spheres" "--config teapot --spp 256"; do
  for i in 1 2; do
    run "wide   $cfg" "$cfg" 3
    FIREWORK_WIDE=0 run "pair   $cfg" "$cfg" 3
    FIREWORK_LIB=$R/firework_amd/lib/variants/lib_scan5.so run "scan5  $cfg" "$cfg" 3
  done
done 2>&1 | tee $OUT/wide_ab.txt
echo "== full-spp parity"
timeout -k 10 600 python tools/full_parity.py --out $OUT/full_parity.jsonl C3_suzanne:512 teapot:64 C5_part2_all:256 C1_random_spheres:64 2> $OUT/full_parity.err | cut -c1-330
