#!/bin/bash
# round 4: suite, full-spp parity of every config, timings, one-shot
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tee $OUT/pytest.log | tail -4; echo "pytest rc=$?"
echo "== full-spp parity"
timeout -k 10 900 python tools/full_parity.py --out $OUT/full_parity.jsonl C5_part2_all:256 C3_suzanne:512 teapot:64 C1_random_spheres:64 C2_cornell_box:1024 C4a_hdri_test:512 C4b_volume_test:512 2> $OUT/full_parity.err | cut -c1-200
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
echo "== timings"
for cfg in "" "--config C3_suzanne" "--config C5_part2_all --spp 256" "--config teapot --spp 256" "--config C1_random_spheres" "--config C4a_hdri_test" "--config C4b_volume_test"; do run "default $cfg" "$cfg" 5; done 2>&1 | tee $OUT/timings.txt
echo "== one-shot"
FIREWORK_TRACE=1 timeout -k 10 200 python tools/oneshot.py 3 2>&1 | grep -v "amdgpu.ids" | tee $OUT/oneshot.txt | grep "rep0\|arena\|enqueue 1\|enqueue [2-9]" | cut -c1-220
