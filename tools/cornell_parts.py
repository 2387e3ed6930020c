"""How k_extend_linear's time splits between cornell's six rectangles and its two rotated boxes (12 more rectangle tests)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from firework_amd import scenes, _lib
for drop in (0, 1, 2):
    s, r = scenes.config("C2_cornell_box", None, None, 256)
    if drop: s.render_objects = s.render_objects[: 8 - drop]
    r.time_kernels(True)
    ds = _lib.DeviceScene(s.to_desc(), 0)
    ds.render(r)
    st = ds.render(r).stats
    print(f"objects {8 - drop}: frame {st['ms_render']:.2f} ms, extend {st['ms_extend']:.2f}, shade {st['ms_shade']:.2f}, rays/sample {st['rays']/st['samples']:.2f}")
    ds.close()
