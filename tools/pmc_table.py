"""Per-kernel summary of a tools/r02_baseline.sh counter set: python tools/pmc_table.py gpurun_out/r02a/pmc_C3_suzanne [out.json]
Joins the SQ / TCC counter passes with the kernel durations of the same passes (kernel_trace.csv).
SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants)."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
def short(n): return n.split("(")[0].replace("void ", "").strip()
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(list)
nl = collections.Counter()
for d in sorted(os.listdir(root)):
    for f in glob.glob(f"{root}/{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("fw::"):
                cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if d == "sqa" and r["Counter_Name"] == "SQ_WAVE_CYCLES": nl[k] += 1
    for f in glob.glob(f"{root}/{d}/**/*kernel_trace.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("fw::"): per[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for k, v in per.items(): dur[k].append(v)
CLK, SIMDS = 2.4e9, 1024
out = {}
for k, c in cnt.items():
    us = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else 0.0
    wc = c.get("SQ_WAVE_CYCLES", 0) * 4
    row = {"us_total(median of passes)": round(us, 1),
           "launches": nl[k],
           "waves": c.get("SQ_WAVES"),
           "avg_waves_per_simd@2.4GHz": round(wc / (us * 1e-6 * CLK * SIMDS), 2) if us else None,
           "wave_time_split": {"waiting(s_waitcnt)": round(c.get("SQ_WAIT_ANY", 0) * 4 / wc, 3) if wc else None,
                               "issue_stall": round(c.get("SQ_WAIT_INST_ANY", 0) * 4 / wc, 3) if wc else None,
                               "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) * 4 / wc, 3) if wc else None},
           "valu_busy_frac@2.4GHz": round(c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (us * 1e-6 * CLK * SIMDS), 3) if us else None,
           "valu_cycles_per_inst": round(c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / c["SQ_INSTS_VALU"], 2) if c.get("SQ_INSTS_VALU") else None,
           "lane_utilisation": round(c.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * c["SQ_INSTS_VALU"]), 3) if c.get("SQ_INSTS_VALU") else None,
           "insts": {n[9:]: c.get(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS")},
           "hbm_read_bytes(FETCH_SIZE*1024*2)": c.get("FETCH_SIZE", 0) * 2048, "hbm_write_bytes": c.get("WRITE_SIZE", 0) * 1024,
           "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3) if c.get("TCC_HIT_sum") else None}
    if us: row["hbm_GBps"] = round((row["hbm_read_bytes(FETCH_SIZE*1024*2)"] + row["hbm_write_bytes"]) / (us * 1e-6) / 1e9, 1)
    out[k] = row
js = json.dumps(out, indent=1)
print(js)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(js)
