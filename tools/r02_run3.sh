TAG=${1:-r02c}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
FIREWORK_TRACE=1 timeout -k 10 200 python3 tools/oneshot.py 10 > $O/oneshot.log 2>&1; echo "oneshot rc=$?" | tee -a $O/summary.txt
grep -c "rep" $O/oneshot.log
