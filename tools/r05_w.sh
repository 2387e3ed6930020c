#!/bin/bash
# round 5: the number of wave queues against whole rounds of resident waves (k_shade holds 5 per SIMD since round 3, the scans 5-7): cornell, hdri
O=$PWD/gpurun_out/$1; mkdir -p $O
run() { FIREWORK_WAVES=$1 timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('waves %-6s %-24s' % ('$1', '$2'), 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k['ms_extend'],2), 'shd', round(k['ms_shade'],2))"; }
for cfg in "" "--config C4a_hdri_test"; do
  for w in 0 20480 25600 30720 35840 40960 51200 0; do run $w "$cfg"; done
done 2>&1 | tee $O/waves.txt
