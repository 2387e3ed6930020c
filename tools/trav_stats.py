"""Lane utilisation of the BVH walks (debug build with -DFW_TRAV_STATS, see tools/README.md):
FIREWORK_LIB=firework_amd/lib/dbg/lib_travstats.so python tools/trav_stats.py C3_suzanne 16"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from firework_amd import scenes, _lib
cfg, spp = sys.argv[1], int(sys.argv[2])
scene, renderer = scenes.config(cfg, None, None, spp)
ds = _lib.DeviceScene(scene.to_desc(), 0)
lib = C.CDLL(os.environ["FIREWORK_LIB"])
out = (C.c_ulonglong * 8)()
lib.fw_debug_trav_stats(out)
renderer.time_kernels(True)
st = ds.render(renderer).stats
assert lib.fw_debug_trav_stats(out) == 0
names = ["TLAS node visits", "TLAS leaf object tests", "BLAS node visits", "BLAS triangle tests"]
rays = st["rays"]
print(f"{cfg} @{spp}spp: {rays} rays ({st['parked_rays']} parked for k_blas), extend {st['ms_extend']:.1f} ms")
if st["parked_rays"]:      # k_blas builds count busy lanes in slots 0..3 (the TLAS of such scenes is scanned, not walked)
    names[0], names[1] = "k_blas busy lanes / node-loop iteration", "k_blas busy lanes / round"
for i, n in enumerate(names):
    useful, slots = out[2 * i], out[2 * i + 1]
    if slots:
        print(f"  {n:24s} {useful / rays:8.2f} per ray, lane utilisation {useful / slots:6.1%}, wave-iterations per 64 rays {slots / rays:8.2f}")
