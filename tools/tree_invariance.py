import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from firework_amd import _lib, scenes
for name, spp in (("C5_part2_all", 8), ("C3_suzanne", 16), ("C1_random_spheres", 64)):
    s, r = scenes.config(name, None, None, spp)
    _lib.set_option("BVH", None)
    a = r.render_full(s)
    _lib.set_option("BVH", "median")
    b = r.render_full(s)
    _lib.set_option("BVH", None)
    print(name, "rays", a.stats["rays"], b.stats["rays"], "identical image:", np.array_equal(a.linear, b.linear), "differing pixels:", int((a.linear != b.linear).any(axis=1).sum()))
