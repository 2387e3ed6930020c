"""Host-side scaling check: scene creation (reference-tree ranks + SAH build + upload) and a render for big meshes."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from firework_amd.api import *
from firework_amd import _lib

def grid_mesh(n, material):
    xs = np.linspace(-4, 4, n, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = (0.4 * np.sin(2 * X) * np.cos(2 * Z)).astype(np.float32)
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).reshape(-1); b = a + 1; c = a + n; d = c + 1
    idx = np.stack([a, b, c, b, d, c], -1).reshape(-1).astype(np.uint32)
    return TriangleMesh.new(verts, idx, None, None, material)

for n in [int(x) for x in os.environ.get("BIG_MESH_N", "101,317,709").split(",")]:      # 20 k, 200 k, 1 M triangles; FIREWORK_TRACE=1 splits scene creation
    sc = Scene.new()
    m = sc.add_material(LambertianMat.with_color((0.7, 0.6, 0.5)))
    mesh = grid_mesh(n, m)
    sc.add_object(RenderObject.new(mesh).position(0.0, 1.0, 0.0))
    sc.add_object(RenderObject.new(XZRect.new(-20.0, 20.0, -20.0, 20.0, -0.5, m)))
    sc.set_environment(SkyEnv.default())
    cam = CameraSettings.default().cam_pos((0.0, 6.0, -12.0)).look_at((0.0, 1.0, 0.0)).field_of_view(40.0)
    r = Renderer.default().width(640).height(360).samples(16).use_bvh(True).camera(cam)
    t0 = time.time(); ds = _lib.DeviceScene(sc.to_desc()); t1 = time.time()
    res = ds.render(r); t2 = time.time()
    st = res.stats
    print(f"tris={mesh.num_tris()} create={t1-t0:.2f}s render={st['ms_render']:.1f}ms rays={st['rays']} Mrays/s={st['rays']/st['ms_render']/1e3:.0f} depth tlas/blas={(st['reserved']>>16)&0x7fff}/{st['reserved']&0xffff} blas_nodes(ref)={st['blas_nodes']}", flush=True)
    ds.close()
