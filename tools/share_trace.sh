export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/share_trace; mkdir -p $R/gpurun_out/share_trace && cd /tmp
SHARE_WORLDS=8 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/share_trace -- python3 $R/tools/share.py > $R/gpurun_out/share_trace/log.txt 2>&1
tail -2 $R/gpurun_out/share_trace/log.txt
