#!/bin/bash
# round 5: (1) queue counts x XCD-contiguous assignment for rank 0's share of a 2-, 4-, 8-rank cornell frame (lib_xcd4k: the assignment from 4 096
# queues on); (2) FW_WIDE_FMA against the subtract-first box arithmetic on the tree configs, GPU suite under it first; (3) C5 at 4096 spp
# against the oracle on a pixel lattice, with a heartbeat (the oracle leg is silent for minutes)
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD; V=$R/firework_amd/lib/variants
echo "== share: queue counts with the XCD-contiguous assignment from 4096 queues on"
for NW in 0 12288 16384 20480 24576 28672; do
  FIREWORK_LIB=$V/lib_xcd4k.so FIREWORK_WAVES=$NW SHARE_WORLDS=1,2,4,8 timeout -k 10 200 python3 tools/share.py 2>/dev/null | grep "^world" | cut -c1-150 | sed "s/^/xcd4k waves=$NW /"
done 2>&1 | tee $OUT/share_xcd.txt
FIREWORK_WAVES=0 SHARE_WORLDS=1,2,4,8 timeout -k 10 200 python3 tools/share.py 2>/dev/null | grep "^world" | cut -c1-150 | sed "s/^/base waves=0 /" | tee -a $OUT/share_xcd.txt
for v in xcd4k base; do L=$V/lib_$v.so; [ $v = base ] && L=$R/firework_amd/lib/libfirework_hip.so; for i in 1 2; do FIREWORK_LIB=$L timeout -k 10 100 python3 bench.py --config C1_random_spheres --steps 10 --warmup 2 --no-cpu-baseline --no-one-shot 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v C1 ms', round(d['ms_per_step'],3))"; done; done | tee -a $OUT/share_xcd.txt
echo "== FW_WIDE_FMA"
bash tools/r05_ab.sh $1 "nofma base" "base" 2 -- "--config C3_suzanne" "--config teapot --spp 128" "--config C5_part2_all --spp 256" "--config C1_random_spheres" "--config C3_suzanne --spp 64"
echo "== C5 @4096 parity"
( timeout -k 10 1000 python3 bench.py --config C5_part2_all --steps 2 --warmup 1 --no-one-shot --parity-seconds 150 > $OUT/c5_4096.json 2> $OUT/c5_4096.err; echo "rc=$?" > $OUT/c5_rc.txt ) &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 45; echo "heartbeat $(date +%s)"; done
cat $OUT/c5_rc.txt; python3 -c "
import json; d=json.load(open('$OUT/c5_4096.json')); print({k: d[k] for k in ('value','ms_per_step')}); print('parity', d['parity'])"
