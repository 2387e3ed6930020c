"""Load balance of tile-dealing schemes: rays per rank (max / mean) for N-rank splits of a config, on one GPU."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from firework_amd import scenes, _lib
from firework_amd.tiles import tile_pixel_ids
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2_cornell_box"
scene, renderer = scenes.config(cfg, None, None, 64)
s = renderer.settings
ds = _lib.DeviceScene(scene.to_desc(), 0)
renderer.time_kernels(True)
for tile in (32, 16, 8):
    for scheme in ("roundrobin", "diagonal", "hash"):
        line = f"{cfg} tile {tile:2d} {scheme:10s}"
        for world in (2, 4, 8):
            rays, ms = [], []
            for r in range(world):
                ids = tile_pixel_ids(s["width"], s["height"], r, world, tile, scheme)
                out = torch.zeros((ids.shape[0], 3), dtype=torch.uint8, device="cuda")
                st = ds.render(renderer, pixel_ids=ids, out_device_ptrs=(out.data_ptr(), None, None))
                st = ds.render(renderer, pixel_ids=ids, out_device_ptrs=(out.data_ptr(), None, None))
                rays.append(st["rays"]); ms.append(st["ms_render"])
            line += f" | N={world}: rays max/mean {max(rays) / np.mean(rays):.3f} time max/mean {max(ms) / np.mean(ms):.3f}"
        print(line, flush=True)
