#!/bin/bash
# round 5: the shares with and without the phase lock (each setting twice in a row)
O=$PWD/gpurun_out/$1; mkdir -p $O
for s in "" "" "FIREWORK_PHASE_LOCK=0" "FIREWORK_PHASE_LOCK=0" "FIREWORK_PHASE_LOCK=1" "FIREWORK_PHASE_LOCK=1"; do
  echo "== ${s:-default}"; env $s timeout -k 10 200 python3 tools/share.py 2>/dev/null
done | tee $O/share_lock.txt
