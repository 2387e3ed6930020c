O=gpurun_out/r02q; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
run() { python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'rg', round(k.get('ms_raygen',0),2), 'frac', round(d['roofline']['frac'],3))"; }
for rep in 1 2 3; do
  run recompute ""; FIREWORK_NO_RECOMPUTE0=1 run load ""
done 2>&1 | tee $O/ab.txt
run recompute "--config C4a_hdri_test --spp 128"; FIREWORK_NO_RECOMPUTE0=1 run load "--config C4a_hdri_test --spp 128"
run recompute "--config C1_random_spheres"; FIREWORK_NO_RECOMPUTE0=1 run load "--config C1_random_spheres"
