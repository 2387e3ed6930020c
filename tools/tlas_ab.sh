# refilling BVH walks (default) vs the chunked k_extend_bvh (FIREWORK_TLAS_REFILL=0), interleaved, each run under a timeout
R=$PWD
run() { timeout -k 10 150 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
unset FIREWORK_TLAS_REFILL; run refill "--config C3_suzanne --spp 64"; run refill "--config C5_part2_all --spp 16"
export FIREWORK_TLAS_REFILL=0; run chunk "--config C3_suzanne --spp 64"; run chunk "--config C5_part2_all --spp 16"
done
