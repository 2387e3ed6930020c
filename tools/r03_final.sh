#!/bin/bash
# Round-3 evidence, part 1: GPU tests, the bench line, every config at full size, the multi-rank rehearsals.   usage: tools/r03_final.sh <tag>
TAG=${1:-r03z}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tee $O/tests.log | tail -3; echo "tests rc=$?" | tee -a $O/summary.txt
timeout -k 10 400 python3 bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt; cut -c1-400 $O/bench.json
timeout -k 10 900 bash tools/configs.sh 2>$O/configs.err | tee $O/configs_full.jsonl; echo "configs rc=$?" | tee -a $O/summary.txt
timeout -k 10 200 python3 bench.py --config teapot --steps 2 --warmup 1 --no-cpu-baseline --no-one-shot 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'config': d['config']['workload'], 'ms_per_frame': round(d['ms_per_step'],2), 'Mrays/s': round(d['value']), 'kernel_ms': d.get('kernel_ms_per_step')}))" | tee $O/teapot.json
timeout -k 10 200 python3 bench.py --gpus 3 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_3ranks_gloo_one_gpu.json 2>$O/gloo.err; echo "gloo rc=$?" | tee -a $O/summary.txt
timeout -k 10 200 python3 bench.py --force-collective --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_rccl_world1.json 2>$O/rccl.err; echo "rccl rc=$?" | tee -a $O/summary.txt
FIREWORK_TRACE=1 timeout -k 10 200 python3 tools/oneshot.py 4 > $O/oneshot.log 2>&1; tail -4 $O/oneshot.log
cat $O/summary.txt
