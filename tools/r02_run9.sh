O=gpurun_out/r02k; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
AB_C3=1 bash tools/ab_bvh.sh 2>&1 | tee -a $O/ab.txt
AB_C3=1 bash tools/ab_bvh.sh 2>&1 | tee -a $O/ab.txt
