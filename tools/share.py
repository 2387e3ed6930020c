"""Per-rank share of a strong-scaled frame on ONE GPU: renders rank 0's tiles of an N-rank split (no gather) and
reports wall time per frame -> the fixed per-frame overhead that limits tile scaling.  python tools/share.py"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from firework_amd import scenes
from firework_amd.tiles import TiledRenderer

scene, renderer = scenes.config("C2_cornell_box", None, None, None)
renderer.time_kernels(os.environ.get("SHARE_NO_TIMING") != "1")      # SHARE_NO_TIMING=1: no per-launch events (ext/shd read 0): the frame as bench.py's timed loop runs it, graph replay included
base = None
for world in [int(x) for x in os.environ.get("SHARE_WORLDS", "1,2,4,8").split(",")]:
    tr = TiledRenderer(scene, renderer, 0, world, 0, dist=None)
    tr.tg.world = 1; tr.tg.collective = False   # no collective: assemble() just scatters the local tiles
    tr.tg.all_ids_dev = [tr.tg.all_ids_dev[0]]
    for _ in range(int(os.environ.get("SHARE_WARMUP", "2"))): tr.render_frame()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(os.environ.get("SHARE_FRAMES", "5"))
    for _ in range(n): tr.render_frame()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / n
    st = tr.last_stats
    if world == 1:
        base = ms
    eff = f"ideal {base / world:.2f} ms, efficiency {base / world / ms:.2%}" if base else "ideal n/a (run world 1 first: SHARE_WORLDS=1,...)"
    print(f"world {world}: {ms:7.2f} ms/frame wall, device {st['ms_render']:.2f} ms, {eff} ext {st['ms_extend']:.2f} shd {st['ms_shade']:.2f} rg {st['ms_raygen']:.2f} acc {st['ms_accumulate']:.2f} batches {st['n_batches']}", flush=True)
    tr.close()
