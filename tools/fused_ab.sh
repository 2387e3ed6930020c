python -m pytest tests -m gpu -x -q 2>&1 | tail -3 &&
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])" &&
FIREWORK_SPLIT=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])" &&
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
