# Round-2 evidence run: GPU tests, the bench line, every config at full size, rocprofv3 passes for C2 / C3 / C5, multi-rank rehearsals.
TAG=${1:-r02z}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/tests.log
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 bash tools/configs.sh > $O/configs_full.jsonl 2>$O/configs.err; echo "configs rc=$?" | tee -a $O/summary.txt
timeout -k 10 200 python3 bench.py --gpus 3 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_3ranks_gloo_one_gpu.json 2>$O/gloo.err; echo "gloo rc=$?" | tee -a $O/summary.txt
timeout -k 10 200 python3 bench.py --force-collective --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot > $O/bench_rccl_world1.json 2>$O/rccl.err; echo "rccl rc=$?" | tee -a $O/summary.txt
bash tools/prof.sh ${TAG}_c2 > $O/prof_c2.log 2>&1; echo "prof c2 rc=$?" | tee -a $O/summary.txt
cd $R; bash tools/prof.sh ${TAG}_c3 --config C3_suzanne --spp 64 > $O/prof_c3.log 2>&1; echo "prof c3 rc=$?" | tee -a $O/summary.txt
cd $R; bash tools/prof.sh ${TAG}_c5 --config C5_part2_all --spp 16 > $O/prof_c5.log 2>&1; echo "prof c5 rc=$?" | tee -a $O/summary.txt
cd $R; FIREWORK_TRACE=1 timeout -k 10 200 python3 tools/oneshot.py 6 > $O/oneshot.log 2>&1
cat $O/summary.txt
