O=gpurun_out/r02m; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
bash tools/ab_bvh.sh 2>&1 | tee -a $O/ab.txt
export FIREWORK_NO_LDS_TRIS=1; echo NO_LDS_TRIS; AB_C3=1 bash tools/ab_bvh.sh 2>&1 | grep lds | tee -a $O/ab.txt; unset FIREWORK_NO_LDS_TRIS
bash tools/ab_bvh.sh 2>&1 | tee -a $O/ab.txt
