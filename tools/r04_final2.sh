#!/bin/bash
# Round-4 evidence, last pass (after the counter summaries are in profiles/): A/B-build suite, the FUSED switch, the bench line, every config, teapot, shares, first calls
set -o pipefail
TAG=${1:-r04z}; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; V=$R/firework_amd/lib/variants
timeout -k 10 600 python3 -m pytest tests -m gpu -q 2>&1 | tee $O/tests.log | tail -3
FIREWORK_LIB=$V/lib_ab.so timeout -k 10 600 python3 -m pytest tests -m gpu -q 2>&1 | tee $O/tests_ab_build.log | tail -3
for sw in "FIREWORK_SHADE_LIST=1" "FIREWORK_FUSED=1" "FIREWORK_NO_SHADE_DEFER=1" "FIREWORK_TLAS_REFILL=0"; do echo "== FIREWORK_LIB=lib_ab.so $sw"; env FIREWORK_LIB=$V/lib_ab.so $sw timeout -k 10 500 python3 -m pytest tests/test_gpu_divergence.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2; done 2>&1 | tee $O/switches_ab.txt
timeout -k 10 400 python3 bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
timeout -k 10 900 bash tools/configs.sh 2>$O/configs.err > $O/configs_full.jsonl; echo "configs rc=$?"
timeout -k 10 200 python3 bench.py --config teapot --steps 2 --warmup 1 --no-cpu-baseline --no-one-shot 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print(json.dumps({'config': d['config']['workload'], 'ms_per_frame': round(d['ms_per_step'],2), 'Mrays/s': round(d['value']), 'kernel_ms': d.get('kernel_ms_per_step'), 'roofline': {k: r.get(k) for k in ('kernel','bound','bound_source','frac')}}))" | tee $O/teapot.json
timeout -k 10 200 python3 bench.py --gpus 3 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_3ranks_gloo_one_gpu.json 2>$O/gloo3.err; echo "gloo3 rc=$?"
timeout -k 10 200 python3 bench.py --gpus 5 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_5ranks_gloo_one_gpu.json 2>$O/gloo5.err; echo "gloo5 rc=$?"
timeout -k 10 200 python3 bench.py --force-collective --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_rccl_world1.json 2>$O/rccl.err; echo "rccl rc=$?"
FIREWORK_TRACE=1 timeout -k 10 200 python3 tools/oneshot.py 4 > $O/oneshot.log 2>&1; grep -v "^\[" $O/oneshot.log | tail -12
for i in 1 2; do timeout -k 10 200 python3 tools/share.py 2>/dev/null; done | tee $O/share.txt
