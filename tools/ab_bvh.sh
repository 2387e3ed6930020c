# A/B of library variants on the BVH configs, each run under its own timeout (AB_C3=1: suzanne only)
R=$PWD
run() { timeout -k 10 150 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', '$2', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for f in $R/firework_amd/lib/variants/lib_*.so; do v=$(basename $f .so); export FIREWORK_LIB=$f
  run $v "--config C3_suzanne --spp 64"; if [ -z "$AB_C3" ]; then run $v "--config C5_part2_all --spp 16"; run $v "--config C1_random_spheres"; fi
done
