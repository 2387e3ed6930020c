"""Second randomized parity sweep: crowded scenes (TLAS depth), several meshes (only the first one a ray meets is
parked, the rest are walked inline), a medium whose boundary is a mesh / cone / cylinder, HDR environment,
image-textured spheres.  python tools/fuzz_many2.py [first_seed] [count]"""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_binding as ob
from firework_amd.api import *
from firework_amd import scenes


def blob_mesh(rng, n_lat, n_lon, material, with_attr):
    """closed lat-long blob with radial noise"""
    verts, uvs = [], []
    for i in range(n_lat + 1):
        th = math.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * math.pi * j / n_lon
            rr = 1.0 + 0.25 * rng.random()
            verts.append([rr * math.sin(th) * math.cos(ph), rr * math.cos(th), rr * math.sin(th) * math.sin(ph)])
            uvs.append([j / n_lon, i / n_lat])
    idx = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
            c, d = a + n_lon, b + n_lon
            idx += [a, b, c, b, d, c]
    verts = np.array(verts, np.float32)
    nrm = verts / np.maximum(np.linalg.norm(verts, axis=1, keepdims=True), 1e-3)
    return TriangleMesh.new(verts, idx, nrm if with_attr else None, np.array(uvs, np.float32) if with_attr else None, material)


def scene(seed):
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))
    sc = Scene.new()
    img = (rng.random((16, 16, 3)) * 255).astype(np.uint8)
    mats = [sc.add_material(LambertianMat.with_color((u(.2, .9), u(.2, .9), u(.2, .9)))), sc.add_material(LambertianMat.new(ImageTexture.new(img))),
            sc.add_material(MetalMat.new((.8, .8, .8), u(0, .3))), sc.add_material(DielectricMat.new(1.5)), sc.add_material(EmissiveMat.with_color((4., 4., 4.)))]
    pick = lambda: int(rng.choice(mats))
    for _ in range(int(rng.integers(50, 200))):
        sc.add_object(RenderObject.new(Sphere.new(u(.1, .4), pick())).position(u(-6, 6), u(.1, 4), u(-6, 6)))
    for k in range(3):
        sc.add_object(RenderObject.new(blob_mesh(rng, 6, 8, pick(), k != 1)).rotate(Rotor3.from_euler_angles(u(-1, 1), u(-1, 1), u(-1, 1)))
                      .position(u(-3, 3), u(1, 3), u(-3, 3)))
    inner = [blob_mesh(rng, 5, 6, mats[0], False), Cone.new(1.0, 1.5, mats[0]), Cylinder.new(0.8, 1.5, mats[0]), Sphere.new(1.0, mats[0])][seed % 4]
    sc.add_volume(RenderObject.new(inner).position(u(-2, 2), 0.5, u(-2, 2)), u(.5, 2), ConstantTexture.new((.8, .8, .9)))
    sc.add_object(RenderObject.new(XZRect.new(-40., 40., -40., 40., -0.21, mats[0])))
    if seed % 2:
        sc.set_environment(HdrEnvironment(scenes.synthetic_hdr(256, 128)))
    else:
        sc.set_environment(SkyEnv.default())
    cam = CameraSettings.default().cam_pos((u(-3, 3), u(3, 6), -14.)).look_at((0., 1.5, 0.)).field_of_view(40.)
    return sc, cam


first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 20
diffs = []
for seed in range(first, first + count):
    sc, cam = scene(seed)
    for bvh in (False, True):
        r = Renderer.default().width(60).height(40).samples(4).use_bvh(bvh).camera(cam).seed(seed)
        g = r.render_full(sc); c = ob.render(sc, r)
        bad = int((g.rgb8 != c.rgb8).any(axis=1).sum())                     # round 4: the reference's output type, bit for bit
        dr = g.stats["rays"] - c.stats["rays"]
        if bad or dr or [int(x) for x in g.stats["rays_per_depth"]] != [int(x) for x in c.stats["rays_per_depth"]]:
            diffs.append((seed, bvh, bad, dr))
print("cases with any difference:", diffs)
print("total", len(diffs), "of", 2 * count)
