#!/bin/bash
# round 5: is this box one of those where k_shade has two speeds from process to process?  Six fresh processes, exclusive kernel times;
# if the spread of k_shade exceeds 5 %, the in-process arena test (tools/r05_m.py) follows in the same call.
O=$PWD/gpurun_out/$1; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-one-shot 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('process $i: ms', round(d['ms_per_step'],2), 'shd', round(k['ms_shade'],2), 'ext', round(k['ms_extend'],2))"; done | tee $O/modes.txt
python3 - <<PY
import re
v=[float(m.group(1)) for m in re.finditer(r"shd ([0-9.]+)", open("$O/modes.txt").read())]
print("k_shade spread", min(v), max(v), "bimodal" if max(v) > 1.05*min(v) else "one speed")
open("$O/bimodal","w").write("1" if max(v) > 1.05*min(v) else "0")
PY
if [ "$(cat $O/bimodal)" = 1 ]; then
  for i in 1 2 3; do echo "== process $i, seven arenas"; timeout -k 10 200 python3 tools/r05_m.py C2_cornell_box 7 2>/dev/null | grep arena; done | tee $O/arenas.txt
fi
