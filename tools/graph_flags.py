import os, sys
sys.path.insert(0, os.getcwd())
from firework_amd import _lib, scenes
_lib.init(0)
for name, w, h, spp in (("C1_random_spheres", 100, 56, 8), ("C3_suzanne", 96, 54, 4)):
    s, r = scenes.config(name, w, h, spp)
    ds = _lib.DeviceScene(s.to_desc(), 0)
    _lib.set_option("GRAPH", "0"); ds.render(r)
    _lib.set_option("GRAPH", "1")
    print(name, [ds.render(r).stats["reserved"] >> 31 for _ in range(5)])
