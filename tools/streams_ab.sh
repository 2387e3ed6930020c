p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"; }
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | p streams1
FIREWORK_STREAMS=2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | p streams2
FIREWORK_STREAMS=4 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | p streams4
done
