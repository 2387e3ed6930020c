"""Where k_exact_paths' time goes (debug build with -DFW_EXACT_PROF, see tools/README.md):
FIREWORK_LIB=firework_amd/lib/dbg/lib_exactprof.so python tools/exact_prof.py C3_suzanne 64"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from firework_amd import scenes, _lib
cfg, spp = sys.argv[1], int(sys.argv[2])
scene, renderer = scenes.config(cfg, None, None, spp)
ds = _lib.DeviceScene(scene.to_desc(), 0)
lib = C.CDLL(os.environ["FIREWORK_LIB"])
out = (C.c_ulonglong * 16)()
ds.render(renderer)
lib.fw_debug_exact_prof(out)
renderer.time_kernels(True)
st = ds.render(renderer).stats
assert lib.fw_debug_exact_prof(out) == 0
paths, segs, nlit, clit, nfast, cfast, cshade, call, cmax, smax, launches = [int(x) for x in out[:11]]
us = lambda c: c / 100.0          # s_memrealtime: 100 MHz
print(f"{cfg} @{spp}spp: {st['rays']} rays, frame {st['ms_render']:.2f} ms; k_exact_paths {launches} launches")
if paths:
    print(f"  {paths} paths left the wavefront ({paths / max(1, st['samples']):.3%} of the samples), {segs / paths:.2f} segments each")
    print(f"  literal walks  {nlit:9d}  {us(clit) / max(1, nlit):8.1f} us each  {us(clit) / paths:8.1f} us per path")
    print(f"  ordinary walks {nfast:9d}  {us(cfast) / max(1, nfast):8.1f} us each  {us(cfast) / paths:8.1f} us per path")
    print(f"  shading        {segs:9d}  {us(cshade) / max(1, segs):8.1f} us each  {us(cshade) / paths:8.1f} us per path")
    print(f"  a path: {us(call) / paths:.1f} us on average, the longest {us(cmax):.0f} us ({smax} segments)")
