"""Randomized GPU-vs-oracle parity sweep (development tool): python tools/fuzz_many.py [first_seed] [count]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_binding as ob
from firework_amd.api import Renderer
import test_gpu_parity as T

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = []
max_rel = 0.0
for seed in range(first, first + count):
    sc, cam = T._random_scene(seed)
    for bvh in (False, True):
        r = Renderer.default().width(60).height(40).samples(6).use_bvh(bvh).camera(cam).seed(seed * 7919)
        g = r.render_full(sc); c = ob.render(sc, r)
        # the reference's output (Vec<Color>: 3 x u8 per pixel) bit for bit, and the rays per depth; the pre-gamma float means are
        # reported, not required: the device multiplies a path's attenuations front to back ((a0 a1) a2) e, the reference's
        # recursion back to front a0 (a1 (a2 e)) (render.rs:24-28) — the same product to within an ulp or two
        bad8 = int((g.rgb8 != c.rgb8).sum())
        dr = [a - b for a, b in zip(g.stats["rays_per_depth"], c.stats["rays_per_depth"])]
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.abs(g.linear - c.linear) / np.maximum(np.abs(c.linear), 1e-6)
        max_rel = max(max_rel, float(np.nanmax(rel)) if rel.size else 0.0)
        if bad8 > 0 or any(dr):
            worst.append((seed, bvh, bad8, dr))
    if (seed - first) % 100 == 99:
        print(f"  ... {seed - first + 1} seeds, {len(worst)} cases with a difference so far", flush=True)
print("cases with any difference (seed, use_bvh, differing u8, rays per depth GPU - oracle):", worst)
print(f"largest relative difference of a pre-gamma mean: {max_rel:.3g}")
print("total", len(worst), "of", 2 * count, "(60x40 @6 spp each, random scenes of tests/test_gpu_parity.py::_random_scene, with and without use_bvh)")
