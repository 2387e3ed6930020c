"""Where the pre-gamma means of GPU and oracle differ most in the random scenes of tools/fuzz_many.py: python tools/fuzz_linear.py [first] [count]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_binding as ob
from firework_amd.api import Renderer
import test_gpu_parity as T
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 100
top = []
for seed in range(first, first + count):
    sc, cam = T._random_scene(seed)
    for bvh in (False, True):
        r = Renderer.default().width(60).height(40).samples(6).use_bvh(bvh).camera(cam).seed(seed * 7919)
        g = r.render_full(sc); c = ob.render(sc, r)
        d = np.abs(g.linear - c.linear); rel = d / np.maximum(np.abs(c.linear), 1e-30)
        rel[np.isnan(rel)] = 0
        i = np.unravel_index(np.argmax(rel), rel.shape)
        top.append((float(rel[i]), seed, bvh, int(i[0]), int(i[1]), float(g.linear[i]), float(c.linear[i])))
top.sort(reverse=True)
for t in top[:12]:
    print("rel %.3g seed %d bvh %s pixel %d channel %d gpu %.9g oracle %.9g" % t)
ulps = []
