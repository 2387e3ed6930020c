# All BASELINE configs at full size on one GPU (parity cases; only C2 is the bench line).
R=$PWD
run() {
  python3 $R/bench.py --config $1 --steps $2 --warmup $3 --no-cpu-baseline --no-one-shot 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print(json.dumps({'config': d['config']['workload'], 'Mrays/s': round(d['value']), 'Msamples/s': round(d['msamples_per_s']), 'ms_per_frame': round(d['ms_per_step'],1), 'rays_per_sample': round(d['rays_per_sample'],2), 'ms_extend': round(k.get('ms_extend',0),1), 'ms_shade': round(k.get('ms_shade',0),1)}))"
}
run C1_random_spheres 3 1
run C2_cornell_box 3 1
run C3_suzanne 2 1
run C4a_hdri_test 2 1
run C4b_volume_test 2 1
run C5_part2_all 1 0
