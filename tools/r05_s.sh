#!/bin/bash
# round 5: GRAPH (a repeating frame replayed as one hipGraph) against plain launches: GPU suite with it, then each setting twice in a row
O=$PWD/gpurun_out/$1; mkdir -p $O
FIREWORK_GRAPH=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tee $O/tests_graph1.log | tail -3
run() { env $1 timeout -k 10 300 python3 bench.py --steps ${3:-50} --warmup 5 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-18s %-40s' % ('$1', '$2'), 'ms', round(d['ms_per_step'],3), 'Mrays/s', round(d['value']))"; }
for cfg in "--config C1_random_spheres --spp 64" "--config C1_random_spheres" "--config C3_suzanne --spp 16" "--config C2_cornell_box --spp 64"; do
  for g in FIREWORK_GRAPH=0 FIREWORK_GRAPH=0 FIREWORK_GRAPH=1 FIREWORK_GRAPH=1; do run $g "$cfg"; done
done 2>&1 | tee $O/graph_ab.txt
for g in FIREWORK_GRAPH=0 FIREWORK_GRAPH=1; do run $g "" 10; done 2>&1 | tee -a $O/graph_ab.txt
for g in FIREWORK_GRAPH=0 FIREWORK_GRAPH=0 FIREWORK_GRAPH=1 FIREWORK_GRAPH=1; do echo "== share $g"; env $g timeout -k 10 200 python3 tools/share.py 2>/dev/null; done 2>&1 | tee $O/share_graph.txt
