R=$PWD
run() {
  timeout -k 10 120 python3 $R/bench.py --config $1 --steps 2 --warmup 1 --no-cpu-baseline $2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1 $2', 'Mrays/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'ext', round(k.get('ms_extend',0),1), 'shd', round(k.get('ms_shade',0),1))"
}
run C3_suzanne && run C5_part2_all "--spp 256" && run C1_random_spheres && run C2_cornell_box
