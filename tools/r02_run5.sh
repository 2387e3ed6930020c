cd $GRAFT_REPO_ROOT; O=gpurun_out/r02g; mkdir -p $O
FIREWORK_LIB=$PWD/firework_amd/lib/dbg/lib_travstats.so timeout -k 10 200 python3 tools/trav_stats.py C3_suzanne 64 2>&1 | grep -v amdgpu | tee $O/trav.txt
for rep in 1 2; do bash tools/ab_bvh.sh 2>&1 | grep -v random | tee -a $O/ab.txt; done
