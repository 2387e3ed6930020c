O=gpurun_out/r02s; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for r in 1 2; do bash tools/ab_bvh.sh 2>&1 | tee -a $O/ab.txt; done
FIREWORK_NO_LDS_TREES=1 bash tools/ab_bvh.sh 2>&1 | grep -v suzanne | sed 's/^/nolds /' | tee -a $O/ab.txt
