CFG=$1; SPP=$2
export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/pmcx; mkdir -p $R/gpurun_out/pmcx && cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmcx/a -- python3 $R/bench.py --config $CFG --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcx/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmcx/b -- python3 $R/bench.py --config $CFG --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcx/b.log 2>&1
tail -1 $R/gpurun_out/pmcx/a.log | cut -c1-200
