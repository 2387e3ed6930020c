# Round-2 baseline on one box: GPU tests, all configs at full size, SQ + HBM counters of the BVH configs.
# usage: bash tools/r02_baseline.sh <tag>
TAG=${1:-r02a}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 bash tools/configs.sh > $O/configs_full.jsonl 2>$O/configs.err; echo "configs rc=$?" | tee -a $O/summary.txt
cd /tmp
pmc() {  # cfg spp name counters...
  local cfg=$1 spp=$2 name=$3; shift 3
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$cfg/$name -- python3 $R/bench.py --config $cfg --spp $spp --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_${cfg}_$name.log 2>&1
  echo "pmc $cfg $name rc=$?" | tee -a $O/summary.txt
}
for c in "C3_suzanne 64" "C5_part2_all 16"; do set -- $c
  pmc $1 $2 sqa SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD &&
  pmc $1 $2 sqb SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES &&
  pmc $1 $2 fetch FETCH_SIZE &&
  pmc $1 $2 write WRITE_SIZE &&
  pmc $1 $2 tcc TCC_HIT_sum TCC_MISS_sum
done
cat $O/summary.txt
