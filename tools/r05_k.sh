# round 5: one or two batches in flight for the linear scans without box lists (hdri, volume), and the phase lock on cornell — each setting twice in
# a row, so that a box whose processes alternate between two k_shade modes (profiles/r05k_lanes_lock.txt, first table) does not decide the comparison
R=$PWD
run() { env $1 timeout -k 10 400 python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('%-44s %-28s' % ('$1', '$2'), 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
mkdir -p gpurun_out/r05k
for cfg in "--config C4a_hdri_test" "--config C4b_volume_test"; do for i in 1 2; do
  run "FIREWORK_STREAMS=1" "$cfg"; run "FIREWORK_STREAMS=1" "$cfg"; run "FIREWORK_STREAMS=2 FIREWORK_PHASE_LOCK=0" "$cfg"; run "FIREWORK_STREAMS=2 FIREWORK_PHASE_LOCK=0" "$cfg"
done; done 2>&1 | tee gpurun_out/r05k/lanes2.txt
for i in 1 2; do run "FIREWORK_PHASE_LOCK=0" ""; run "FIREWORK_PHASE_LOCK=0" ""; run "FIREWORK_PHASE_LOCK=1" ""; run "FIREWORK_PHASE_LOCK=1" ""; done 2>&1 | tee -a gpurun_out/r05k/lanes2.txt
