"""The floor of C4a's k_shade traffic, from the per-dispatch counters of a tools/prof.sh run (FETCH_SIZE and WRITE_SIZE passes):
per segment, the bytes the counters saw against the queue streams that segment moves, and what is left per path that ENDS in the
segment (HdrEnvironment lookup of a 16-byte texel + the 16-byte radiance record).
    python3 tools/hdri_floor.py gpurun_out/r05z_c4a profiles/r05z_full_parity.jsonl C4a_hdri_test > profiles/r05z_c4a_floor.txt"""
import csv, glob, json, os, sys
d, parity, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
rays = None
for l in open(parity):
    if l.startswith("{"):
        j = json.loads(l)
        if j["config"] == cfg:
            rays = j["rays_per_depth_gpu"]
def per_dispatch(sub, scale):
    rows = []
    for f in sorted(glob.glob(f"{d}/{sub}/*/*counter_collection.csv"), key=os.path.getmtime)[-1:]:      # gpurun merges runs additively: the newest
        for r in csv.DictReader(open(f)):
            if "k_shade" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024 * scale))
    rows.sort()
    return [b for _, b in rows]
rd = per_dispatch("fetch", 2.0)      # gfx950: FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md; scripts/summarize_prof.py)
wr = per_dispatch("write", 1.0)
n_seg = len(rays)
n_batches = len(rd) // n_seg          # of all the frames the profiled command rendered (timed, exclusive pass, counting pass ...)
n_frames = 0
for f in sorted(glob.glob(f"{d}/fetch/*/*kernel_trace.csv"), key=os.path.getmtime)[-1:]:
    n_frames += sum(1 for r in csv.DictReader(open(f)) if "k_resolve" in r["Kernel_Name"])
print(f"# {cfg}: {len(rd)} k_shade dispatches = {n_frames} frames x {n_batches // n_frames} batches x {n_seg} segments (STREAMS=1); bytes per frame = sums over its batches, averaged over the frames")
print(f"# streams = rays in x (ray_a 16 + ray_b 8 + state 8 + hit 8) + rays out x (ray_a 16 + ray_b 8 + state 8); segment 0 reads no ray_b/state")
print("segment  rays_in      rays_out     ended        PMC_read_MB PMC_write_MB  streams_MB  left_per_ended_B  (read part / write part)")
tot_left = tot_ended = 0
for s in range(n_seg):
    r_in = rays[s]; r_out = rays[s + 1] if s + 1 < n_seg else 0
    ended = r_in - r_out
    R = sum(rd[b * n_seg + s] for b in range(n_batches)) / n_frames; W = sum(wr[b * n_seg + s] for b in range(n_batches)) / n_frames
    s_rd = r_in * (24 if s == 0 else 40); s_wr = r_out * 32
    left_r, left_w = (R - s_rd) / max(1, ended), (W - s_wr) / max(1, ended)
    tot_left += (R - s_rd) + (W - s_wr); tot_ended += ended
    print(f"{s:7d}  {r_in:11d}  {r_out:11d}  {ended:11d}  {R/1e6:10.1f}  {W/1e6:10.1f}  {(s_rd+s_wr)/1e6:10.1f}  {left_r+left_w:8.1f}  ({left_r:.1f} / {left_w:.1f})")
print(f"# per ended path, frame average: {tot_left / tot_ended:.1f} B beyond the queue streams; the algorithmic count allows 32 (texel 16 + record 16)")
