"""Randomized check of k_extend_linear_defer (development tool): rooms that end in 1-3 boxes, GPU with the box lists vs GPU
without (FIREWORK_NO_DEFER) vs the oracle.  python tools/fuzz_defer.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_binding as ob
from firework_amd import _lib
from firework_amd.api import (CameraSettings, DielectricMat, EmissiveMat, LambertianMat, MetalMat, Rect3d, RenderObject, Renderer,
                              Rotor3, Scene, Sphere, XYRect, XZRect, YZRect)

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))
    sc = Scene.new()
    mats = [sc.add_material(LambertianMat.with_color((u(.2, .9), u(.2, .9), u(.2, .9)))) for _ in range(3)]
    mats += [sc.add_material(MetalMat.new((0.8, 0.8, 0.9), u(0, 0.5))), sc.add_material(DielectricMat.new(u(1.3, 1.7)))]
    light = sc.add_material(EmissiveMat.with_color((u(4, 12),) * 3))
    pick = lambda: int(rng.choice(mats))
    s = u(3, 6)
    sc.add_object(RenderObject.new(XZRect.new(-s, s, -s, s, 0.0, pick())))
    sc.add_object(RenderObject.new(XZRect.new(-s, s, -s, s, s, pick())).flip_normals())
    sc.add_object(RenderObject.new(XZRect.new(-1, 1, -1, 1, s - 0.01, light)).flip_normals())
    if rng.random() < 0.7:
        sc.add_object(RenderObject.new(XYRect.new(-s, s, 0, s, s, pick())).flip_normals())
    if rng.random() < 0.7:
        sc.add_object(RenderObject.new(YZRect.new(0, s, -s, s, -s, pick())))
    for _ in range(int(rng.integers(0, 3))):
        sc.add_object(RenderObject.new(Sphere.new(u(0.3, 1.0), pick())).position(u(-2, 2), u(0.3, 2), u(-2, 2)))
    for b in range(int(rng.integers(1, 4))):
        ro = RenderObject.new(Rect3d.with_size((u(0.5, 2.5), u(0.5, 3.0), u(0.5, 2.5)), pick()))
        k = int(rng.integers(0, 3))
        ro = ro.rotate(Rotor3.identity() if k == 0 else Rotor3.from_rotation_xz(u(-3, 3)) if k == 1 else Rotor3.from_euler_angles(u(-1, 1), u(-1, 1), u(-1, 1)))
        ro = ro.position(u(-2, 1), 0.0 if rng.random() < 0.5 else u(-0.5, 1.5), u(-2, 1))
        if rng.random() < 0.25:
            ro = ro.flip_normals()
        sc.add_object(ro)
    cam = CameraSettings.default().cam_pos((u(-1, 1), u(1, 3), -3 * s)).look_at((0.0, s / 2, 0.0)).field_of_view(u(30, 50))
    r = Renderer.default().width(64).height(48).samples(8).use_bvh(False).camera(cam).seed(seed * 31)
    _lib.set_option("NO_DEFER", None)
    g = r.render_full(sc)
    _lib.set_option("NO_DEFER", "1")
    p = r.render_full(sc)
    _lib.set_option("NO_DEFER", None)
    c = ob.render(sc, r)
    same_gpu = np.array_equal(g.linear, p.linear) and g.stats["rays_per_depth"] == p.stats["rays_per_depth"]
    n_bad = int((g.rgb8 != c.rgb8).any(axis=1).sum())            # round 4: the reference's output type bit for bit
    if not same_gpu or n_bad or [int(x) for x in g.stats["rays_per_depth"]] != [int(x) for x in c.stats["rays_per_depth"]]:
        bad.append((seed, same_gpu, n_bad, g.stats["rays"] - c.stats["rays"]))
print("cases with any difference:", bad)
print("total", len(bad), "of", count)
