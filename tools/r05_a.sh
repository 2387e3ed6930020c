#!/bin/bash
# round 5, first pass (measurement only, HEAD kernels): section statistics of the wide walks and k_shade (lib_phase.so), scene creation split
# for big meshes, a kernel trace of rank 0's share of a 1- and 4-rank cornell frame, the bench line.  Output: gpurun_out/$1/
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD; V=$R/firework_amd/lib/variants
echo "== phase stats"; 
FIREWORK_LIB=$V/lib_phase.so FIREWORK_STREAMS=1 timeout -k 10 400 python3 tools/phase_stats.py C3_suzanne:64 teapot:32 C5_part2_all:16 C1_random_spheres:64 C2_cornell_box:128 C4a_hdri_test:64 C4b_volume_test:64 > $OUT/phase.txt 2> $OUT/phase.err; echo "rc=$?"; tail -3 $OUT/phase.err
echo "== big meshes"
FIREWORK_TRACE=1 BIG_MESH_N=150,317,709 timeout -k 10 400 python3 tools/big_mesh.py > $OUT/big_mesh.txt 2> $OUT/big_mesh.err; echo "rc=$?"
grep -h "tris=\|meshes (" $OUT/big_mesh.txt $OUT/big_mesh.err | cut -c1-260
echo "== share trace"
export TMPDIR=/tmp; cd /tmp
for W in 1 4; do
  SHARE_WORLDS=$W timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/share_w$W -- python3 $R/tools/share.py > $OUT/share_log_$W.txt 2>&1; echo "world $W rc=$?"; grep "^world" $OUT/share_log_$W.txt
done
cd $R
python3 - <<PY
import csv, glob, collections
for W in (1, 4):
    fs = glob.glob("$OUT/share_w%d/**/*kernel_trace.csv" % W, recursive=True)
    if not fs: print("no trace", W); continue
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows: by[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
    print("world", W, "launches", len(rows), "span ms", (t1 - t0) / 1e6)
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])): print("   %-60s n=%5d sum %9.2f ms  mean %8.1f us  max %8.1f us" % (k, len(v), sum(v) / 1e3, sum(v) / len(v), max(v)))
PY
echo "== bench"
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; cut -c1-400 $OUT/bench.json
