#!/bin/bash
# Round-3 evidence, part 2: rocprofv3 passes (kernel stats; FETCH_SIZE, WRITE_SIZE, SQ sets, TCC — separate --pmc passes) for one config.
#   usage: tools/r03_final_prof.sh <tag> <c2|c3|c5>
TAG=$1; W=$2; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
case $W in
  c2) bash tools/prof.sh ${TAG}_c2 2>&1 | tee $O/prof_c2.log ;;
  c3) bash tools/prof.sh ${TAG}_c3 --config C3_suzanne --spp 64 2>&1 | tee $O/prof_c3.log ;;
  c5) bash tools/prof.sh ${TAG}_c5 --config C5_part2_all --spp 16 2>&1 | tee $O/prof_c5.log ;;
esac
