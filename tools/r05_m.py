"""Round 5: does k_shade's two-speed behaviour follow the path arena's placement?  One process, the same scene and frame; between
renders the arena is re-reserved a little larger (fw_init allocates the new one first, so it lands somewhere else).
    python3 tools/r05_m.py [config] [n_arenas]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from firework_amd import _lib, scenes
from firework_amd._lib import DeviceScene

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2_cornell_box"
n_arenas = int(sys.argv[2]) if len(sys.argv) > 2 else 6
GB = 1 << 30
_lib.init(0, 24 * GB)
scene, renderer = scenes.config(cfg, None, None, None)
renderer.time_kernels(True)
ds = DeviceScene(scene if hasattr(scene, "ptr") else scene.to_desc(), 0)
for a in range(n_arenas):
    if a:
        _lib.init(0, (24 + 2 * a) * GB)
    row = []
    for r in range(4):
        st = ds.render(renderer).stats
        row.append((round(st["ms_shade"], 2), round(st["ms_extend"], 2), round(st["ms_render"], 2)))
    print(f"arena {24 + 2 * a} GB: (shd, ext, render) x4 = {row}", flush=True)
