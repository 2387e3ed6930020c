export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02j; rm -rf $O; mkdir -p $O
run() { timeout -k 10 150 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
export FIREWORK_DYNQ=1
for sw in 5120 7168 10240; do export FIREWORK_STREAM_WAVES=$sw; run dyn_sw$sw "--config C3_suzanne --spp 64"; done 2>&1 | tee $O/sweep.txt
export FIREWORK_STREAM_WAVES=7168
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_dyn -- python3 $R/bench.py --config C3_suzanne --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-one-shot > $O/trace_dyn.log 2>&1
echo done
