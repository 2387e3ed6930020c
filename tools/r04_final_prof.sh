#!/bin/bash
# Round-4 evidence: rocprofv3 passes (kernel stats; FETCH_SIZE, WRITE_SIZE, SQ sets, TCC — separate --pmc passes) of every config at the
# size configs.sh runs it (C5 at 256 spp).   usage: tools/r04_final_prof.sh <tag> c1 c2 ...
TAG=$1; shift; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for W in "$@"; do
  case $W in
    c1) bash tools/prof.sh ${TAG}_c1 --config C1_random_spheres 2>&1 | tee $O/prof_c1.log ;;
    c2) bash tools/prof.sh ${TAG}_c2 2>&1 | tee $O/prof_c2.log ;;
    c3) bash tools/prof.sh ${TAG}_c3 --config C3_suzanne 2>&1 | tee $O/prof_c3.log ;;
    c4a) bash tools/prof.sh ${TAG}_c4a --config C4a_hdri_test 2>&1 | tee $O/prof_c4a.log ;;
    c4b) bash tools/prof.sh ${TAG}_c4b --config C4b_volume_test 2>&1 | tee $O/prof_c4b.log ;;
    c5) bash tools/prof.sh ${TAG}_c5 --config C5_part2_all --spp 256 2>&1 | tee $O/prof_c5.log ;;
    teapot) bash tools/prof.sh ${TAG}_teapot --config teapot 2>&1 | tee $O/prof_teapot.log ;;
  esac || exit 1
done
