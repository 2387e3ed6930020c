"""One seed of tools/fuzz_many.py in detail: the pixels whose u8 differs, both sides' pre-gamma means, the scene's materials / textures /
shapes (development tool).  python tools/fuzz_seed.py <seed>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_binding as ob
from firework_amd.api import Renderer
from firework_amd import _lib
import test_gpu_parity as T

seed = int(sys.argv[1])
sc, cam = T._random_scene(seed)
print("materials:", [type(m).__name__ for m in sc.materials])
print("textures:", sorted({type(getattr(m, "texture", None)).__name__ for m in sc.materials}))
for bvh in (False, True):
    for chain in (None, "1"):
        _lib.set_option("NO_CHAIN", chain)
        r = Renderer.default().width(60).height(40).samples(6).use_bvh(bvh).camera(cam).seed(seed * 7919)
        g = r.render_full(sc); c = ob.render(sc, r)
        bad = np.nonzero((g.rgb8 != c.rgb8).any(axis=1))[0]
        print("bvh", bvh, "NO_CHAIN", chain, "chain bytes", g.stats["bytes_shade"], "u8 diffs at", bad.tolist(), "rays equal", [int(x) for x in g.stats["rays_per_depth"]] == [int(x) for x in c.stats["rays_per_depth"]])
        for i in bad[:3]:
            print("   pixel", int(i), "gpu u8", g.rgb8[i], "cpu u8", c.rgb8[i], "gpu linear", g.linear[i], "cpu linear", c.linear[i], "rel", np.abs(g.linear[i] - c.linear[i]) / np.maximum(np.abs(c.linear[i]), 1e-9))
_lib.set_option("NO_CHAIN", None)
# round 5: the same seed with the attenuations multiplied back to front (option EXACT_PRODUCT): the pre-gamma means must be the oracle's bits
_lib.set_option("EXACT_PRODUCT", "1")
for bvh in (False, True):
    r = Renderer.default().width(60).height(40).samples(6).use_bvh(bvh).camera(cam).seed(seed * 7919)
    g = r.render_full(sc); c = ob.render(sc, r)
    print("EXACT_PRODUCT bvh", bvh, "u8 diffs", int((g.rgb8 != c.rgb8).sum()), "linear equal bit for bit", bool(np.array_equal(g.linear, c.linear.astype(np.float32))),
          "max abs diff", float(np.abs(g.linear - c.linear).max()), "rays equal", [int(x) for x in g.stats["rays_per_depth"]] == [int(x) for x in c.stats["rays_per_depth"]])
_lib.set_option("EXACT_PRODUCT", None)
