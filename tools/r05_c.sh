#!/bin/bash
# round 5: (1) k_shade ablations + XCD swizzle, interleaved; (2) per-segment kernel times of one exclusive cornell frame (rocprofv3 trace, STREAMS=1)
# with the rays per depth next to them; (3) the A/B build's error-word test; (4) EXACT_PRODUCT cost on C5@256 / C1 / earth
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
bash tools/r05_ab.sh $1 "base xcd nodep nopix" "-" 2 -- "" "--config C4a_hdri_test"
echo "== per-segment trace (exclusive)"
export TMPDIR=/tmp; cd /tmp
FIREWORK_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/seg -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-shot > $OUT/seg_bench.json 2> $OUT/seg_bench.err; echo "rc=$?"
cd $R
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/seg/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fw::", "")[:24] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_raygen")]
i0 = idx[-2]      # the last frame's first batch
seg_e, seg_s = [], []
for r, n in list(zip(rows, names))[i0:idx[-1]]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n.startswith("k_extend"): seg_e.append(d)
    if n.startswith("k_shade"): seg_s.append(d)
print("extend us per segment:", [round(x) for x in seg_e])
print("shade  us per segment:", [round(x) for x in seg_s])
PY
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from firework_amd import scenes
s, r = scenes.config("C2_cornell_box")
st = r.render_full(s).stats
print("rays per depth (whole frame, two batches):", [int(x) for x in st["rays_per_depth"]])
PY
echo "== error word (A/B build)"
FIREWORK_LIB=$R/firework_amd/lib/variants/lib_ab.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "error_word or wide_node or chain_state" 2>&1 | tail -4
echo "== EXACT_PRODUCT cost"
for cfg in "--config C5_part2_all --spp 256" "--config C1_random_spheres" "--config earth"; do for i in 1 2; do
  for e in 0 1; do FIREWORK_EXACT_PRODUCT=$e timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('EXACT_PRODUCT=$e %-34s' % '$cfg', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; done
done; done 2>&1 | tee $OUT/exact_product.txt
