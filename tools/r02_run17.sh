O=gpurun_out/r02t; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
run() { timeout -k 10 150 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for r in 1 2 3; do
 run wg2 "--config C3_suzanne --spp 64"; FIREWORK_LDS_WG2=0 run wg1 "--config C3_suzanne --spp 64"
 FIREWORK_STREAMS=1 run wg2_s1 "--config C3_suzanne --spp 64"; FIREWORK_STREAMS=1 FIREWORK_LDS_WG2=0 run wg1_s1 "--config C3_suzanne --spp 64"
done
run wg2 "--config C3_suzanne"; FIREWORK_LDS_WG2=0 run wg1 "--config C3_suzanne"
