O=gpurun_out/r02o; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
bash tools/ab_bvh.sh 2>&1 | grep -v random | tee -a $O/ab.txt
bash tools/ab_bvh.sh 2>&1 | grep -v random | tee -a $O/ab.txt
