export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/c3trace; mkdir -p $R/gpurun_out/c3trace && cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3trace -- python3 $R/bench.py --config C3_suzanne --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/c3trace/log.txt 2>&1
cd $R; FIREWORK_LIB=$PWD/firework_amd/lib/dbg/lib_travstats.so timeout -k 10 100 python tools/trav_stats.py C3_suzanne 16 2>&1 | grep -v amdgpu
