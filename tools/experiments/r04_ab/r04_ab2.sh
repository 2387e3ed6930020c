#!/bin/bash
# round 4: what exact gating (reference leaf-node boxes) and the SOFT class cost; the chain state on cornell; the last diverging C5 path
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tee $OUT/pytest.log | tail -12; echo "pytest rc=$?"
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
echo "== gating / soft class A/B"
for cfg in "--config C3_suzanne" "--config C5_part2_all --spp 256" "--config teapot --spp 256"; do
  for i in 1 2; do
    run "default        $cfg" "$cfg" 3
    FIREWORK_OWN_BOXES=1 run "own_boxes      $cfg" "$cfg" 3
    FIREWORK_OWN_BOXES=1 FIREWORK_SOFT_SHEAR_LOG2=0 run "own+nosoft     $cfg" "$cfg" 3
    FIREWORK_OWN_BOXES=1 FIREWORK_SOFT_SHEAR_LOG2=0 FIREWORK_WIDE=0 run "own+nosoft+pair $cfg" "$cfg" 3
  done
done 2>&1 | tee $OUT/gating_ab.txt
echo "== cornell chain state"
for i in 1 2 3; do run "chain   " "" 6; FIREWORK_NO_CHAIN=1 run "product " "" 6; done 2>&1 | tee $OUT/chain_ab.txt
for cfg in "--config C4a_hdri_test" "--config C4b_volume_test"; do for i in 1 2; do run "chain   $cfg" "$cfg" 4; FIREWORK_NO_CHAIN=1 run "product $cfg" "$cfg" 4; done; done 2>&1 | tee -a $OUT/chain_ab.txt
echo "== diverge C5"
timeout -k 10 300 python tools/diverge.py C5_part2_all 1920 1080 256 --pixels 845940 --out $OUT/diverge_C5.json > $OUT/C5.txt 2>&1; tail -c 300 $OUT/C5.txt
