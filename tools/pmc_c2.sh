export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/pmc2; mkdir -p $R/gpurun_out/pmc2 && cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/a -- python3 $R/bench.py --spp 128 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc2/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/b -- python3 $R/bench.py --spp 128 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc2/b.log 2>&1
echo done $?
