#!/bin/bash
set -o pipefail
# Round-5 evidence from the final kernel sources.   usage: tools/r04_final.sh <tag> <part>
#   1: GPU tests (default build, then the -DFW_AB=1 build), the bench line, every config at full size, teapot, multi-rank rehearsals, first calls, shares
#   2: full-sample-count parity of every config against the oracle (tools/full_parity.py) + the randomized sweep
#   3: the parity tests under the switches that select other kernels
TAG=${1:-r05z}; PART=${2:-1}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; V=$R/firework_amd/lib/variants
case $PART in
1)
  timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tee $O/tests.log | tail -3; echo "tests rc=$?" | tee -a $O/summary.txt
  FIREWORK_LIB=$V/lib_ab.so timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tee $O/tests_ab_build.log | tail -3; echo "tests (FW_AB build) rc=$?" | tee -a $O/summary.txt
  timeout -k 10 400 python3 bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt; cut -c1-400 $O/bench.json
  timeout -k 10 900 bash tools/configs.sh 2>$O/configs.err | tee $O/configs_full.jsonl; echo "configs rc=$?" | tee -a $O/summary.txt
  timeout -k 10 200 python3 bench.py --config teapot --steps 2 --warmup 1 --no-cpu-baseline --no-one-shot 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'config': d['config']['workload'], 'ms_per_frame': round(d['ms_per_step'],2), 'Mrays/s': round(d['value']), 'kernel_ms': d.get('kernel_ms_per_step')}))" | tee $O/teapot.json
  timeout -k 10 200 python3 bench.py --gpus 3 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_3ranks_gloo_one_gpu.json 2>$O/gloo3.err; echo "gloo3 rc=$?" | tee -a $O/summary.txt
  timeout -k 10 200 python3 bench.py --gpus 5 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_5ranks_gloo_one_gpu.json 2>$O/gloo5.err; echo "gloo5 rc=$?" | tee -a $O/summary.txt
  timeout -k 10 200 python3 bench.py --force-collective --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_rccl_world1.json 2>$O/rccl.err; echo "rccl rc=$?" | tee -a $O/summary.txt
  FIREWORK_TRACE=1 timeout -k 10 200 python3 tools/oneshot.py 4 > $O/oneshot.log 2>&1; grep -v "^\[" $O/oneshot.log | tail -12
  for i in 1 2; do timeout -k 10 200 python3 tools/share.py 2>/dev/null; done | tee $O/share.txt
  cat $O/summary.txt ;;
1b)
  FIREWORK_LIB=$V/lib_ab.so timeout -k 10 600 python3 -m pytest tests -m gpu -q 2>&1 | tee $O/tests_ab_build.log | tail -3; echo "tests (FW_AB build) rc=$?" | tee -a $O/summary.txt ;;
2)
  rm -f $O/full_parity.jsonl
  ( while true; do sleep 60; echo "heartbeat $(date +%s)"; done ) & HB=$!
  timeout -k 10 1100 python3 tools/full_parity.py --out $O/full_parity.jsonl | cut -c1-260; echo "full parity rc=$?" | tee -a $O/summary.txt
  timeout -k 10 500 python3 tools/fuzz_many.py 20000 600 2>&1 | tail -3 | tee $O/fuzz.txt
  kill $HB ;;
3)
  for sw in "FIREWORK_STREAMS=1" "FIREWORK_NO_LDS_TREES=1" "FIREWORK_WIDE=0" "FIREWORK_WIDE=q8" "FIREWORK_BVH=median" "FIREWORK_NO_LDS_TRIS=1" "FIREWORK_STREAMS=4" "FIREWORK_NO_HOIST=1" "FIREWORK_NO_CHAIN=1" "FIREWORK_NO_DEFER=1" "FIREWORK_EXACT_ALL=1" "FIREWORK_EXACT_PRODUCT=1" "FIREWORK_PHASE_LOCK=1" "FIREWORK_GRAPH=1" "FIREWORK_NO_SHORT_RAYS=1" \
            "FIREWORK_LIB=$V/lib_ab.so FIREWORK_TLAS_REFILL=0" "FIREWORK_LIB=$V/lib_ab.so FIREWORK_SHADE_LIST=1" "FIREWORK_LIB=$V/lib_ab.so FIREWORK_FUSED=1"; do
    echo "== ${sw//$V\//}"
    env $sw timeout -k 10 500 python3 -m pytest tests/test_gpu_divergence.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 || { echo "FAILED under $sw"; exit 1; }
  done 2>&1 | tee $O/switches.txt ;;
3b)
  for sw in "FIREWORK_LIB=$V/lib_ab.so FIREWORK_TLAS_REFILL=0" "FIREWORK_LIB=$V/lib_ab.so FIREWORK_SHADE_LIST=1" "FIREWORK_LIB=$V/lib_ab.so FIREWORK_FUSED=1"; do
    echo "== ${sw//$V\//}"
    env $sw timeout -k 10 500 python3 -m pytest tests/test_gpu_divergence.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 || { echo "FAILED under $sw"; exit 1; }
  done 2>&1 | tee $O/switches_b.txt ;;
4)
  # the last pass, on the committed build with its counter summaries under profiles/: tests, the bench line (roofline.bound and traffic from
  # the counters), the shares, C1 repeated, C5 at its own 4096 spp against the oracle on a pixel lattice (heartbeat: the oracle leg is silent)
  timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tee $O/tests.log | tail -3; echo "tests rc=$?" | tee -a $O/summary4.txt
  timeout -k 10 400 python3 bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary4.txt; cut -c1-300 $O/bench.json
  for i in 1 2; do timeout -k 10 200 python3 tools/share.py 2>/dev/null; done | tee $O/share.txt
  for i in 1 2 3; do timeout -k 10 100 python3 bench.py --config C1_random_spheres --spp 64 --steps 20 --warmup 3 --no-cpu-baseline --no-one-shot 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C1 @64 ms', round(d['ms_per_step'],3), 'Mrays/s', round(d['value']))"; done | tee $O/c1.txt
  ( timeout -k 10 1000 python3 bench.py --config C5_part2_all --steps 2 --warmup 1 --no-one-shot --parity-seconds 150 > $O/c5_4096.json 2> $O/c5_4096.err; echo "rc=$?" > $O/c5_rc.txt ) &
  PID=$!
  while kill -0 $PID 2>/dev/null; do sleep 45; echo "heartbeat $(date +%s)"; done
  cat $O/c5_rc.txt | tee -a $O/summary4.txt
  python3 -c "
import json; d=json.loads(open('$O/c5_4096.json').read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step')}); print('parity', d['parity'])" ;;
esac
