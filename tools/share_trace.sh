# per-launch durations of rank 0's share for SHARE_WORLDS (default 1 and 8): where the small share loses efficiency
export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/share_trace; mkdir -p $R/gpurun_out/share_trace && cd /tmp
for W in ${SHARE_WORLDS_LIST:-1 8}; do
SHARE_WORLDS=$W rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/share_trace/w$W -- python3 $R/tools/share.py > $R/gpurun_out/share_trace/log_$W.txt 2>&1
done
echo done
