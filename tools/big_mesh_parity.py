import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
from firework_amd.api import *
from oracle import oracle_binding as ob
import importlib.util
spec = importlib.util.spec_from_file_location("bm", "tools/big_mesh.py")
src = open("tools/big_mesh.py").read().split("for n in (101")[0]
ns = {"__file__": os.path.abspath("tools/big_mesh.py")}; exec(src, ns)
sc = Scene.new()
m = sc.add_material(LambertianMat.with_color((0.7, 0.6, 0.5)))
mesh = ns["grid_mesh"](150, m)
sc.add_object(RenderObject.new(mesh).position(0.0, 1.0, 0.0))
sc.add_object(RenderObject.new(XZRect.new(-20.0, 20.0, -20.0, 20.0, -0.5, m)))
sc.set_environment(SkyEnv.default())
cam = CameraSettings.default().cam_pos((0.0, 6.0, -12.0)).look_at((0.0, 1.0, 0.0)).field_of_view(40.0)
r = Renderer.default().width(160).height(90).samples(4).use_bvh(True).camera(cam)
g = r.render_full(sc); c = ob.render(sc, r)
print("44k-triangle mesh: rays", g.stats["rays"], c.stats["rays"], "identical:", np.array_equal(g.linear, c.linear), "max diff", float(np.abs(g.linear - c.linear).max()))
