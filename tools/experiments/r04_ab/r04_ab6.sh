#!/bin/bash
# round 4: kernel traces of teapot / suzanne, cornell by wave-queue count, the last C5 path, one-shot trace
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
R=$PWD
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tee $OUT/pytest.log | tail -4; echo "pytest rc=$?"
for cfg in "teapot 64" "C3_suzanne 64"; do set -- $cfg
  (cd /tmp && FIREWORK_STREAMS=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$1 -- python3 $R/bench.py --config $1 --spp $2 --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot > $OUT/trace_$1.log 2>&1)
  echo "== kernel stats $1 @$2 (one batch in flight)"; f=$(find $OUT/trace_$1 -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-5 | cut -c1-150
done
run() { timeout -k 10 300 python3 $R/bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'excl', round(d['schedule']['exclusive_pass_ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'Mrays/s', round(d['value']))"; }
echo "== cornell by wave queues"
for i in 1 2 3; do run "default " "" 8; FIREWORK_WAVES=12288 run "w12288  " "" 8; FIREWORK_WAVES=16384 run "w16384  " "" 8; FIREWORK_WAVES=28672 run "w28672  " "" 8; done 2>&1 | tee $OUT/cornell_waves.txt
for cfg in "--config C4a_hdri_test" "--config C3_suzanne" "--config teapot --spp 256"; do run "default $cfg" "$cfg" 4; done 2>&1 | tee $OUT/misc.txt
echo "== one-shot"
FIREWORK_TRACE=1 timeout -k 10 200 python tools/oneshot.py 3 2>&1 | grep -v "amdgpu.ids" | tee $OUT/oneshot.txt | grep "rep0\|arena" | cut -c1-220
echo "== diverge C5 @256, whole frame"
timeout -k 10 400 python tools/diverge.py C5_part2_all 1920 1080 256 --max-pixels 3 --out $OUT/diverge_C5.json > $OUT/C5.txt 2>&1; tail -c 400 $OUT/C5.txt
