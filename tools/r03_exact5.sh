#!/bin/bash
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; R=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/pytest.log | tail -4
run() { timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>$OUT/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))" || tail -5 $OUT/err.txt; }
echo "== exact walk on / off"
for cfg in "--config C3_suzanne --spp 64" "--config C3_suzanne" "--config C5_part2_all --spp 16" "--config teapot --spp 32" "--config C1_random_spheres"; do
  for i in 1 2; do run "exact   $cfg" "$cfg"; FIREWORK_NO_EXACT=1 run "noexact $cfg" "$cfg"; done
done 2>&1 | tee $OUT/exact_ab.txt
export TMPDIR=/tmp; cd /tmp
rm -rf $OUT/prof; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --config C3_suzanne --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:44].ljust(46), r['Calls'].rjust(5), '%10.1f us avg'%(float(r['AverageNs'])/1e3), '%8.2f ms total'%(float(r['TotalDurationNs'])/1e6), 'max %.0f us'%(float(r['MaxNs'])/1e3))
PY
