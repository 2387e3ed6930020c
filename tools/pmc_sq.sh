# SQ / SQC counters for the cornell bench: tools/pmc_sq.sh <tag> [spp]   (env assignments are inherited)
TAG=$1; SPP=${2:-256}
export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/pmcs_$TAG; mkdir -p $R/gpurun_out/pmcs_$TAG && cd /tmp
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_$TAG/a -- python3 $R/bench.py --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcs_$TAG/a.log 2>&1 &&
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_$TAG/b -- python3 $R/bench.py --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcs_$TAG/b.log 2>&1
echo pmc_sq $TAG exit $?
