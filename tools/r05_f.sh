#!/bin/bash
# round 5: GPU suite on the product build; wave-queue counts swept for rank 0's share of a 4- and an 8-rank cornell frame; the bench line
# (fw_init / cold start both ways / native cpu baseline); C5 at its own 4096 spp against the oracle on a pixel lattice (--parity-seconds 150)
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest.log | tail -4; echo "pytest rc=$?"
echo "== share waves sweep"
for W in 4 8; do for NW in 0 8192 10240 12288 14336 16384 20480 28672; do
  FIREWORK_WAVES=$NW SHARE_WORLDS=$W timeout -k 10 200 python3 tools/share.py 2>/dev/null | grep "^world" | sed "s/^/waves=$NW /"
done; done 2>&1 | tee $OUT/share_waves.txt
echo "== bench line"
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print({k: d[k] for k in ('value','ms_per_step')}); print('roofline', {k: d['roofline'][k] for k in ('bound','bound_source','frac','kernel')}); print('cold', {k: v for k, v in d['one_shot_cold'].items() if k.startswith('ms_')}, 'lazy', {k: v for k, v in d['one_shot_cold'].get('without_fw_init', {}).items() if k.startswith('ms_')}); print('cpu', {k: d['cpu_baseline'][k] for k in ('value','cores','build')}, d['cpu_baseline'].get('checker_build', {}).get('value')); print('parity', d['parity'])"
echo "== C5 @4096 parity"
timeout -k 10 900 python3 bench.py --config C5_part2_all --steps 2 --warmup 1 --no-one-shot --parity-seconds 150 > $OUT/c5_4096.json 2> $OUT/c5_4096.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('$OUT/c5_4096.json')); print({k: d[k] for k in ('value','ms_per_step')}); print('parity', d['parity'])"
