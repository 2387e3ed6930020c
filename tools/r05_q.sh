#!/bin/bash
# round 5: 16-byte camera rays for pinhole cameras with a +0.0 position coordinate (hdri_test, volume_test: x = 0.0): NO_SHORT_RAYS=1 (24 B, as before)
# against the default, each twice in a row; the parity files under the default first
O=$PWD/gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tee $O/tests.log | tail -3
run() { env $1 timeout -k 10 400 python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('%-26s %-28s' % ('${1:-default}', '$2'), 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k['ms_extend'],2), 'shd', round(k['ms_shade'],2), 'rg', round(k['ms_raygen'],2), 'Mrays/s', round(d['value']))"; }
for cfg in "--config C4a_hdri_test" "--config C4b_volume_test"; do
  for s in "FIREWORK_NO_SHORT_RAYS=1" "FIREWORK_NO_SHORT_RAYS=1" "" ""; do run "$s" "$cfg"; done
done 2>&1 | tee $O/short_rays_ab.txt
