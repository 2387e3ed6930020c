#!/bin/bash
# round 4: per-kernel split of the tree configs (one batch in flight), kernel trace only.  usage: bash tools/r04_trace.sh <tag> [config:spp ...]
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O && cd /tmp
export FIREWORK_STREAMS=1
for cs in "$@"; do
  c=${cs%%:*}; spp=${cs##*:}
  rm -rf /tmp/tr_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$c -- python3 $R/bench.py --config $c --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot > $O/trace_$c.log 2>&1 || exit 1
  f=$(find /tmp/tr_$c -name "*kernel_stats.csv" | head -1)
  echo "== $c @ $spp" | tee -a $O/kernel_split.txt
  python3 - "$f" <<'PY' | tee -a $O/kernel_split.txt
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:9]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void fw::", "")
    print(f"{n[:46]:46s} calls {int(r['Calls']):5d}  avg_us {float(r['AverageNs'])/1e3:9.1f}  total_ms {float(r['TotalDurationNs'])/1e6:8.2f}  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
done
