import sys, os
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_binding as ob
from firework_amd.api import Renderer
import test_gpu_parity as T
seed = int(sys.argv[1])
sc, cam = T._random_scene(seed)
for bvh in (False, True):
    r = Renderer.default().width(60).height(40).samples(6).use_bvh(bvh).camera(cam).seed(seed * 7919)
    g = r.render_full(sc); c = ob.render(sc, r)
    scale = np.maximum(np.abs(c.linear), 1e-3)
    badm = (np.abs(g.linear - c.linear) > 2e-4 * scale + 1e-6).any(axis=1)
    print(os.environ.get("FIREWORK_LIB", "default")[-14:], "bvh", bvh, "bad", int(badm.sum()), "rays", g.stats["rays"], c.stats["rays"], "idx", np.nonzero(badm)[0][:5], g.linear[badm][:2], c.linear[badm][:2])
