"""round 5: the leaf-list variant under the experiment build's checks: small suzanne and teapot frames, status and error word"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from firework_amd import _lib, scenes
from firework_amd._lib import DeviceScene
_lib.init(0)
for cfg, w, h, spp in (("C3_suzanne", 160, 90, 4), ("teapot", 160, 90, 4), ("C3_suzanne", 640, 360, 16)):
    scene, renderer = scenes.config(cfg, w, h, spp)
    ds = DeviceScene(scene if hasattr(scene, "ptr") else scene.to_desc(), 0)
    try:
        r = ds.render(renderer)
        print(cfg, w, h, spp, "ok rays", r.stats["rays"], flush=True)
    except Exception as e:
        print(cfg, w, h, spp, "FAILED:", str(e)[:300], flush=True)
