# batches in flight on separate HIP streams (FIREWORK_STREAMS=n), interleaved, cornell bench
run() { python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'Mrays/s', round(d['value']))"; }
for rep in 1 2; do for n in 1 2 3 4; do FIREWORK_STREAMS=$n run streams$n; done; done
python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks gloo one gpu', round(d['ms_per_step'],2))"
python3 bench.py --gpus 4 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-one-shot 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4 ranks gloo one gpu', round(d['ms_per_step'],2))"
