# All BASELINE configs at full size on one GPU (parity cases; only C2 is the bench line): one JSON line per config with the frame
# time of the library's own schedule, the per-kernel times of the exclusive pass and the roofline objects (profiles/*_configs_full.jsonl).
R=$PWD
run() {
  timeout -k 10 900 python3 $R/bench.py --config $1 --steps $2 --warmup $3 --no-cpu-baseline --no-one-shot 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); r=d.get('roofline',{}); o=r.get('other_kernel',{})
print(json.dumps({'config': d['config']['workload'], 'Mrays/s': round(d['value']), 'Msamples/s': round(d['msamples_per_s']), 'ms_per_frame': round(d['ms_per_step'],2),
  'rays_per_sample': round(d['rays_per_sample'],2), 'schedule': d.get('schedule',{}).get('timed_loop'), 'exclusive_ms_per_frame': round(d.get('schedule',{}).get('exclusive_pass_ms_per_step',0),2),
  'exclusive_kernel_ms': {a: round(b,2) for a,b in k.items()},
  'roofline': {'kernel': r.get('kernel'), 'bound': r.get('bound'), 'bound_source': r.get('bound_source'), 'counters': r.get('counters'), 'pmc_MB_per_launch_of_its_longest_kernel': (round(r['traffic'] / 1e6, 1) if r.get('traffic') else None), 'algorithmic_MB_per_launch': (round(r['algorithmic_bytes_per_launch'] / 1e6, 1) if r.get('algorithmic_bytes_per_launch') else None), 'achieved_GBps': round(r.get('achieved',0)), 'frac_of_8TBps': round(r.get('frac',0),3), 'bytes_per_ray': round(r.get('bytes_per_ray',0),1), 'avg_launch_us': round(r.get('avg_launch_us',0),1)},
  'other_kernel': {'kernel': o.get('kernel'), 'bound': o.get('bound'), 'bound_source': o.get('bound_source'), 'counters': o.get('counters'), 'achieved_GBps': round(o.get('achieved',0)), 'frac_of_8TBps': round(o.get('frac', o.get('frac_hbm',0)),3), 'avg_launch_us': round(o.get('avg_launch_us',0),1)},
  'roofline_frame_layout_frac': round(d.get('roofline_frame',{}).get('layout',{}).get('frac',0),3), 'device': {a: b for a,b in d.get('device',{}).items() if a in ('name','sclk_mhz','mclk_mhz','copy_GBps')}}))"
}
run C1_random_spheres 50 5      # (a frame this short is replayed as a graph from its third repetition on: GRAPH)
run C2_cornell_box 3 1
run C3_suzanne 2 1
run C4a_hdri_test 2 1
run C4b_volume_test 2 1
run C5_part2_all 2 1
