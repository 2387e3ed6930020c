"""Sum rocprofv3 counter_collection.csv files per kernel: python tools/pmc_read.py gpurun_out/pmcs_fused"""
import csv, glob, collections, os, sys
root = sys.argv[1]
tot = collections.defaultdict(dict)
for d in sorted(os.listdir(root)):
    fs = sorted(glob.glob(f'{root}/{d}/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:24]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in agg.items():
        if 'fw::' in k: tot[k].update(v)
for k, v in tot.items():
    print(k)
    for a, b in sorted(v.items()): print(f'    {a:28s} {b:.4g}')
