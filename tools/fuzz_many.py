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
for seed in range(first, first + count):
    sc, cam = T._random_scene(seed)
    for bvh in (False, True):
        r = Renderer.default().width(60).height(40).samples(6).use_bvh(bvh).camera(cam).seed(seed * 7919)
        g = r.render_full(sc); c = ob.render(sc, r)
        scale = np.maximum(np.abs(c.linear), 1e-3)
        bad = int((np.abs(g.linear - c.linear) > 2e-4 * scale + 1e-6).any(axis=1).sum())
        dr = g.stats["rays"] - c.stats["rays"]
        if bad > 0 or dr != 0:
            worst.append((seed, bvh, bad, dr))
print("cases with any difference:", worst)
print("total", len(worst), "of", 2 * count)
