#!/usr/bin/env python3
"""Turns rocprofv3 output directories (gpurun_out/prof/{trace,fetch,write}) into the small summaries
committed under profiles/:  <tag>_kernel_stats.csv (copy of rocprofv3 --kernel-trace --stats) and
<tag>_pmc.json (per-kernel HBM bytes from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes).

Unit / correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counter values are KB (x1024 bytes);
on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced reads, so the read side is
doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
    tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
    os.makedirs("profiles", exist_ok=True)
    newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1:]   # gpurun merges runs additively
    ks = newest(f"{src}/trace/*/*_kernel_stats.csv")
    if ks:
        shutil.copyfile(ks[0], f"profiles/{tag}_kernel_stats.csv")
    out = {"command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
           "note": "bytes = Counter_Value * 1024; read side doubled (gfx950 FETCH_SIZE = 1/2 of wide coalesced reads)",
           "kernels": {}}
    for cname, key, mul in (("fetch", "read_bytes", 2.0 * 1024), ("write", "write_bytes", 1024.0)):
        fs = newest(f"{src}/{cname}/*/*_counter_collection.csv")
        if not fs:
            continue
        tot, cnt = collections.defaultdict(float), collections.Counter()
        for row in csv.DictReader(open(fs[0])):
            k = short(row["Kernel_Name"])
            if not k.startswith("fw::"):
                continue
            tot[k] += float(row["Counter_Value"]) * mul
            cnt[k] += 1
        for k in tot:
            d = out["kernels"].setdefault(k, {})
            d[key] = tot[k]
            d["launches"] = cnt[k]
    for k, d in out["kernels"].items():
        d["hbm_bytes_per_launch"] = (d.get("read_bytes", 0.0) + d.get("write_bytes", 0.0)) / max(1, d["launches"])
    json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
