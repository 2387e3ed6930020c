TAG=${1:-r02b}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
FIREWORK_TRACE=1 timeout -k 10 200 python3 tools/oneshot.py 10 > $O/oneshot.log 2>&1; echo "oneshot rc=$?" | tee -a $O/summary.txt
