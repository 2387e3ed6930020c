# rocprofv3 evidence for profiles/: kernel trace + separate PMC passes.  usage: bash tools/prof.sh <dir tag> [bench args...]
#   e.g.  bash tools/prof.sh prof_c2                      (the bench config)
#         bash tools/prof.sh prof_c3 --config C3_suzanne --spp 64
# then:   python3 scripts/summarize_prof.py gpurun_out/<dir tag> <profiles tag> "<workload>"
TAG=$1; shift
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O && cd /tmp
B="--no-cpu-baseline --no-one-shot"
# one batch in flight in every frame of these passes (round 3: the library's own schedule overlaps two batches, whose launches
# carry half a frame each): per-launch averages then mean the same thing as bench.py's exclusive pass, which the roofline quotes
export FIREWORK_STREAMS=1
(cd $R && python3 -c "import bench; print(bench.kernel_source_sha())") > $O/source_sha.txt     # the build these passes ran
pass() { local name=$1; shift; local ctr="$1"; shift
  if [ -n "$ctr" ]; then timeout -k 10 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -- python3 $R/bench.py "$@" --steps 1 --warmup 0 $B > $O/$name.log 2>&1
  else timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" --steps 3 --warmup 1 $B > $O/$name.log 2>&1; fi
  echo "prof $TAG $name rc=$?"; }
pass trace "" "$@" &&
pass fetch "FETCH_SIZE" "$@" &&
pass write "WRITE_SIZE" "$@" &&
pass sqa "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD" "$@" &&
pass sqb "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "$@" &&
pass tcc "TCC_HIT_sum TCC_MISS_sum" "$@"
tail -1 $O/trace.log | cut -c1-400
