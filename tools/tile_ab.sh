for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile16', d['ms_per_step'], d['kernel_ms_per_step'])"
FW_TILE=32 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile32', d['ms_per_step'], d['kernel_ms_per_step'])"
done
