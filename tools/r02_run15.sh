O=gpurun_out/r02r; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
AB_FULL= bash tools/ab.sh 2>&1 | tee $O/ab.txt
for f in firework_amd/lib/variants/lib_*.so; do v=$(basename $f .so); for cfg in "--config C4b_volume_test --spp 128" "--config C4a_hdri_test --spp 128"; do
FIREWORK_LIB=$PWD/$f python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-one-shot $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$v', '$cfg', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; done; done 2>&1 | tee -a $O/ab.txt
