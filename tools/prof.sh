# rocprofv3 evidence for profiles/: kernel trace + separate PMC passes (FETCH_SIZE, WRITE_SIZE)
export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof && cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof/write.log 2>&1
echo prof exit $?
tail -1 $R/gpurun_out/prof/trace.log | cut -c1-300
