"""Paths with NaN rays (the third class of the exact walk): pixels whose pre-gamma mean is NaN, then the sample.
python tools/find_nan_paths.py C3_suzanne 64   ->  (pixel, sample) pairs with the device's and the oracle's path length"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FIREWORK_NO_ZERO_SKIP"] = "1"          # every path deposits: accum.w is its length in segments
import numpy as np
from firework_amd import _lib, scenes
from oracle import oracle_binding as ob
name, spp = sys.argv[1], int(sys.argv[2])
scene, renderer = scenes.config(name, samples=spp)
frame = renderer.render_full(scene)
pix = np.nonzero(np.isnan(frame.linear).any(axis=1))[0]
print(f"{name} @{spp} spp: {len(pix)} pixels with a NaN mean: {pix.tolist()[:40]}")
one = scenes.config(name, samples=1)[1]
sd = scene.to_desc()
ds = _lib.DeviceScene(sd)
found = []
for p in pix[:12]:
    for s in range(spp):
        accum = np.zeros((1, 4), np.float32)
        ds.render_progressive(one, s, accum, np.array([p], np.uint32))
        if np.isnan(accum[0, :3]).any():
            segs, colour = ob.trace_path(sd, one, int(p), s)
            found.append((int(p), s))
            print(f"  pixel {p} sample {s}: device length {int(accum[0, 3])} colour {accum[0, :3]}; oracle length {int(segs[:, 15].sum())} colour {colour}")
print("CASES entry:", found)
