#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/prof.sh (gpurun_out/<dir>/{trace,fetch,write,sqa,sqb,tcc}) into the summaries
committed under profiles/:

  <tag>_kernel_stats.csv  copy of rocprofv3 --kernel-trace --stats
  <tag>_pmc.json          per-kernel HBM bytes from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes
  <tag>_sq.json           per-kernel SQ summary (wave-time split, VALU busy fraction, lane utilisation, instruction mix)
                          and the `bound` bench.py reports for the kernel — the larger of two utilisations, each relative
                          to what this chip was MEASURED to sustain, when it is >= 0.7:
                            hbm         PMC bytes / time over 6.0 TB/s (= 0.75 of the 8 TB/s peak: what a kernel that only
                                        copies wave-private queues reaches — tools/membench.hip, and k_shade with its shading
                                        replaced by a move)
                            valu_issue  SQ_INSTS_VALU x 3.7 cycles over the SIMD cycles of the launch (tools/issuebench.hip,
                                        eight waves resident: v_fmac 3.0-3.6, min/cmp/cndmask/div_fixup 4.5, v_mad_u64_u32 5.1
                                        cycles per wave64 instruction at the nominal 2.4 GHz; 3.7 is what the rectangle scan
                                        runs at).  `valu_pipe_util` is the same count at the guide's 2 cycles per instruction
                            latency     neither reaches 0.7: waves parked in s_waitcnt behind dependent loads

usage: scripts/summarize_prof.py <dir> <tag> "<workload string of bench.py's config.workload>"

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KB (x1024); on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads, so the read side is doubled; SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*
count quad-cycles summed over waves.  Both json files carry `source_sha` = sha256 of the kernel sources they were taken
from (bench.py quotes them only when it matches the build it runs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402

CLK, SIMDS, HBM_PEAK = 2.4e9, 1024, 8.0e12


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def newest(pat):
    return sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1:]   # gpurun merges runs additively


def counters(src, sub):
    tot, launches = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    fs = newest(f"{src}/{sub}/**/*_counter_collection.csv")
    if not fs:
        return tot, launches, {}
    first = None
    for row in csv.DictReader(open(fs[0])):
        k = short(row["Kernel_Name"])
        if not k.startswith("fw::"):
            continue
        first = first or row["Counter_Name"]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == first:
            launches[k] += 1
    dur = collections.defaultdict(float)
    for f in newest(f"{src}/{sub}/**/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("fw::"):
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return tot, launches, dur


def main():
    src, tag, workload = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else None)
    os.makedirs(f"{ROOT}/profiles", exist_ok=True)
    sha_file = os.path.join(src, "source_sha.txt")         # written by tools/prof.sh on the GPU box: the build that was profiled
    sha = open(sha_file).read().strip() if os.path.exists(sha_file) else kernel_source_sha()
    ks = newest(f"{src}/trace/**/*_kernel_stats.csv")
    if ks:
        shutil.copyfile(ks[0], f"{ROOT}/profiles/{tag}_kernel_stats.csv")
    pmc = {"command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py [--config ..] --steps 1 --warmup 0 --no-cpu-baseline --no-one-shot",
           "note": "bytes = Counter_Value * 1024; read side doubled (gfx950 FETCH_SIZE = 1/2 of wide coalesced reads)",
           "source_sha": sha, "workload": workload, "kernels": {}}
    durs = {}
    for sub, key, mul in (("fetch", "read_bytes", 2.0 * 1024), ("write", "write_bytes", 1024.0)):
        tot, launches, dur = counters(src, sub)
        for k in tot:
            d = pmc["kernels"].setdefault(k, {})
            d[key] = sum(tot[k].values()) * mul
            d["launches"] = launches[k]
            durs.setdefault(k, []).append(dur.get(k, 0.0))
    for k, d in pmc["kernels"].items():
        d["hbm_bytes_per_launch"] = (d.get("read_bytes", 0.0) + d.get("write_bytes", 0.0)) / max(1, d["launches"])
        us = sum(durs[k]) / max(1, len(durs[k]))
        d["us_total_in_these_passes"] = round(us, 1)
        if us:
            d["hbm_GBps"] = round((d.get("read_bytes", 0.0) + d.get("write_bytes", 0.0)) / (us * 1e-6) / 1e9, 1)
    if pmc["kernels"]:
        json.dump(pmc, open(f"{ROOT}/profiles/{tag}_pmc.json", "w"), indent=1)

    c = collections.defaultdict(dict)
    dur_sq = collections.defaultdict(list)
    launches_sq = {}
    for sub in ("sqa", "sqb", "tcc"):
        tot, launches, dur = counters(src, sub)
        for k in tot:
            c[k].update(tot[k])
            launches_sq[k] = launches[k]
            if dur.get(k):
                dur_sq[k].append(dur[k])
    sq = {"command": "rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python3 bench.py ... (tools/prof.sh; two SQ passes + one TCC pass)",
          "note": "SQ cycle counters are quad-cycles summed over waves; busy fractions assume the 2.4 GHz peak clock (the chip runs lower under load, so they are lower bounds)",
          "source_sha": sha, "workload": workload, "kernels": {}}
    for k, v in c.items():
        us = sum(dur_sq[k]) / max(1, len(dur_sq[k]))
        wc = v.get("SQ_WAVE_CYCLES", 0.0) * 4
        simd_cycles = us * 1e-6 * CLK * SIMDS
        row = {"us_total": round(us, 1), "launches": launches_sq.get(k), "waves": v.get("SQ_WAVES"),
               "avg_waves_per_simd": round(wc / simd_cycles, 2) if us else None,
               "wave_time_split": {"waiting_s_waitcnt": round(v.get("SQ_WAIT_ANY", 0) * 4 / wc, 3) if wc else None,
                                   "issue_stall": round(v.get("SQ_WAIT_INST_ANY", 0) * 4 / wc, 3) if wc else None,
                                   "issuing": round(v.get("SQ_ACTIVE_INST_ANY", 0) * 4 / wc, 3) if wc else None},
               "valu_pipe_util": round(v.get("SQ_INSTS_VALU", 0) * 2 / simd_cycles, 3) if us else None,      # 2 cycles per wave64 VALU instruction on a SIMD-32
               "instr_per_simd_cycle": round(sum(v.get(n, 0) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS")) / simd_cycles, 3) if us else None,
               "valu_cycles_per_inst": round(v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / v["SQ_INSTS_VALU"], 2) if v.get("SQ_INSTS_VALU") else None,
               "lane_utilisation": round(v.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * v["SQ_INSTS_VALU"]), 3) if v.get("SQ_INSTS_VALU") else None,
               "insts": {n[9:]: v.get(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS") if n in v},
               "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3) if v.get("TCC_HIT_sum") else None}
        hb = pmc["kernels"].get(k, {}).get("hbm_GBps")
        row["hbm_frac_of_8TBps"] = round(hb * 1e9 / HBM_PEAK, 3) if hb else None
        row["valu_issue_util"] = round(v.get("SQ_INSTS_VALU", 0) * 3.7 / simd_cycles, 3) if us else None      # at the measured 3.7 cycles per instruction
        row["hbm_util_of_streaming"] = round((row["hbm_frac_of_8TBps"] or 0.0) / 0.75, 3)
        vu, hu = row["valu_issue_util"] or 0.0, row["hbm_util_of_streaming"]
        row["bound"] = "latency" if max(vu, hu) < 0.7 else ("hbm" if hu >= vu else "valu_issue")
        sq["kernels"][k] = row
    if sq["kernels"]:
        json.dump(sq, open(f"{ROOT}/profiles/{tag}_sq.json", "w"), indent=1)
    print(json.dumps({"pmc": pmc, "sq": sq}, indent=1))


if __name__ == "__main__":
    main()
