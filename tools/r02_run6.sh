O=gpurun_out/r02h; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for c in "C3_suzanne 64" "C5_part2_all 16"; do set -- $c
FIREWORK_LIB=$PWD/firework_amd/lib/dbg/lib_travstats.so timeout -k 10 200 python3 tools/trav_stats.py $1 $2 2>&1 | grep -v amdgpu | tee -a $O/trav.txt
done
for rep in 1 2; do bash tools/ab_bvh.sh 2>&1 | tee -a $O/ab.txt; done
