TAG=${1:-r02f}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
bash tools/prof.sh ${TAG}_c2 > $O/prof_c2.log 2>&1; echo "prof c2 rc=$?"
cd $R
for c in "C3_suzanne 64" "C5_part2_all 16"; do set -- $c
FIREWORK_LIB=$R/firework_amd/lib/dbg/lib_travstats.so timeout -k 10 200 python3 tools/trav_stats.py $1 $2 2>&1 | grep -v amdgpu | tee -a $O/trav.txt
done
