# A/B of library variants in firework_amd/lib/variants/ (FIREWORK_LIB), interleaved on one box.  AB_FULL=1 adds the BVH configs.
R=$PWD
run() {
  python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2), 'rg', round(k.get('ms_raygen',0),2), 'acc', round(k.get('ms_accumulate',0),2))"
}
for rep in 1 2; do for f in $R/firework_amd/lib/variants/lib_*.so; do v=$(basename $f .so); FIREWORK_LIB=$f run $v; done; done
if [ -n "$AB_FULL" ]; then
for f in $R/firework_amd/lib/variants/lib_*.so; do v=$(basename $f .so); FIREWORK_LIB=$f run $v "--config C3_suzanne --spp 64"; FIREWORK_LIB=$f run $v "--config C5_part2_all --spp 16"; FIREWORK_LIB=$f run $v "--config C4a_hdri_test --spp 128"; done
fi
if [ -n "$AB_NOLDS" ]; then FIREWORK_NO_LDS_TABLES=1 FIREWORK_LIB=$R/firework_amd/lib/variants/lib_base.so run base_nolds; fi
if [ -n "$AB_TEST" ]; then timeout -k 10 300 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3; fi
