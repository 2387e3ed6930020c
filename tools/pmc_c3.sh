export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/pmc3; mkdir -p $R/gpurun_out/pmc3 && cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc3/a -- python3 $R/bench.py --config C3_suzanne --spp 64 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc3/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc3/b -- python3 $R/bench.py --config C3_suzanne --spp 64 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc3/b.log 2>&1
echo done $?
