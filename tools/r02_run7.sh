export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r02i; rm -rf $O; mkdir -p $O
run() { timeout -k 10 150 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for sw in 5120 7168 10240 15360 20480 40960; do export FIREWORK_STREAM_WAVES=$sw
  run sw$sw "--config C3_suzanne --spp 64"; run sw$sw "--config C5_part2_all --spp 16"; done 2>&1 | tee $O/sweep.txt
unset FIREWORK_STREAM_WAVES
cd /tmp
for v in base stream; do
FIREWORK_LIB=$R/firework_amd/lib/variants/lib_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -- python3 $R/bench.py --config C3_suzanne --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-one-shot > $O/trace_$v.log 2>&1
FIREWORK_LIB=$R/firework_amd/lib/variants/lib_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace5_$v -- python3 $R/bench.py --config C5_part2_all --spp 16 --steps 1 --warmup 1 --no-cpu-baseline --no-one-shot > $O/trace5_$v.log 2>&1
done; echo traces done
