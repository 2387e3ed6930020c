#!/bin/bash
# round 5: a rank's share of a cornell frame as bench.py's timed loop runs it (no per-launch events): graph replay and batches in flight
O=$PWD/gpurun_out/$1; mkdir -p $O
for s in "FIREWORK_GRAPH=0" "FIREWORK_GRAPH=0" "" "" "FIREWORK_STREAMS=1" "FIREWORK_STREAMS=1" "FIREWORK_STREAMS=1 FIREWORK_GRAPH=0"; do
  echo "== ${s:-default}"; env $s SHARE_NO_TIMING=1 SHARE_WARMUP=4 SHARE_FRAMES=20 SHARE_WORLDS=1,2,4,8 timeout -k 10 200 python3 tools/share.py 2>/dev/null | cut -c1-95
done | tee $O/share_notiming.txt
