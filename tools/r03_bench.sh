#!/bin/bash
# round 3: suite, the bench line (two lanes + exclusive pass), lane A/B on cornell, the Infinity-Cache batch experiment, kernel A/Bs
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest.log | tail -5; echo "pytest rc=$?"
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print({k: d.get(k) for k in ('value','ms_per_step','schedule','device','one_shot_cold')}); print(d['roofline']['frac'], d['roofline']['avg_launch_us'], d['kernel_ms_per_step'], d['parity']['rays_equal'], d['parity']['u8_diffs'])"
echo "== cornell lanes A/B"
for i in 1 2 3; do for n in 1 2; do FIREWORK_STREAMS=$n timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$n ms', round(d['ms_per_step'],2))"; done; done 2>&1 | tee $OUT/lanes_ab.txt
echo "== Infinity-Cache-sized batches (paths per batch x streams)"
for ppb in 2097152 8388608 33554432; do for n in 2 4; do FIREWORK_STREAMS=$n FIREWORK_PATHS_PER_BATCH=$ppb timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ppb=$ppb streams=$n ms', round(d['ms_per_step'],2))"; done; done 2>&1 | tee $OUT/mall_batches.txt
R=$PWD
run() { timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
echo "== k_shade list (FIREWORK_NO_SHADE_DEFER=1 is round 2's kernel): part2 @16, hdri @64, random_spheres, volume @64, cornell @128"
for cfg in "--config C5_part2_all --spp 16" "--config C4a_hdri_test --spp 64" "--config C1_random_spheres" "--config C4b_volume_test --spp 64" "--spp 128"; do
  for i in 1 2; do run "list   $cfg" "$cfg"; FIREWORK_NO_SHADE_DEFER=1 run "inline $cfg" "$cfg"; done
done 2>&1 | tee $OUT/shade_ab.txt
