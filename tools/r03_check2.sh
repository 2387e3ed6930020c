#!/bin/bash
# GPU suite + the multi-rank rehearsals (overlapped gather) + cornell lanes 1..4 + random_spheres after the small-frame rule
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest.log | tail -4
timeout -k 10 200 python3 bench.py --gpus 3 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $OUT/bench_3ranks_gloo_one_gpu.json 2>$OUT/gloo.err; echo "gloo rc=$?"; cut -c1-300 $OUT/bench_3ranks_gloo_one_gpu.json
timeout -k 10 200 python3 bench.py --force-collective --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $OUT/bench_rccl_world1.json 2>$OUT/rccl.err; echo "rccl rc=$?"; cut -c1-300 $OUT/bench_rccl_world1.json; tail -3 $OUT/rccl.err
run() { timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>$OUT/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms', round(d['ms_per_step'],2), 'parity', d.get('parity',{}).get('rays_equal'))" || tail -5 $OUT/err.txt; }
for s in 2 3 4 2 3; do FIREWORK_STREAMS=$s run "cornell streams=$s" ""; done 2>&1 | tee $OUT/cornell_lanes.txt
for i in 1 2; do run "random_spheres" "--config C1_random_spheres"; FIREWORK_STREAMS=2 run "random_spheres streams=2" "--config C1_random_spheres"; done 2>&1 | tee $OUT/rs.txt
