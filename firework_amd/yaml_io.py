"""Reads and writes the reference's YAML scene format (serde_yaml + typetag; SURVEY §5 "Serialisation").

Tags: `object_type` (src/serde_compat.rs:22), `material` (src/material.rs:9), `texture` (src/texture.rs:7),
`environment` (src/environment.rs:5); `Rotor3` as `{s, bv: {xy, xz, yz}}` (serde_compat.rs:6-20); `ImageTexture`
as a path under `value` (texture.rs:251-278).  Field names of the shapes that have no committed YAML in the
reference (Rect3d, ConstantMedium, Checker/Metal/Dielectric/Isotropic) follow their struct definitions
(rect3d.rs:10-14, volume.rs:11-15, material.rs:78-81,110-112,184-186).
"""
import os

import numpy as np
import yaml

from . import api as A


class UnsupportedShape(ValueError):
    pass


def _v3(d):
    return (float(d["x"]), float(d["y"]), float(d["z"]))


def _v3d(v):
    return {"x": float(v[0]), "y": float(v[1]), "z": float(v[2])}


# ------------------------------------------------------------------------------------------- load
def _texture(d, base_dir):
    k = d["texture"]
    if k == "ConstantTexture":
        return A.ConstantTexture(_v3(d["color"]))
    if k == "CheckerTexture":
        return A.CheckerTexture(_texture(d["odd"], base_dir), _texture(d["even"], base_dir), float(d["scale"]))
    if k == "PerlinNoiseTexture":
        return A.PerlinNoiseTexture(float(d["scale"]))
    if k == "TurbulenceTexture":
        return A.TurbulenceTexture(int(d["depth"]), float(d["scale"]))
    if k == "MarbleTexture":
        return A.MarbleTexture(int(d["depth"]), float(d["scale"]))
    if k == "ImageTexture":
        # the reference opens the path relative to the process CWD (image::open, texture.rs:288); also try
        # next to the scene file and one level up (scenes/*.yml name images that live in the repo root)
        path = d["value"]
        for cand in (path, os.path.join(base_dir, path), os.path.join(os.path.dirname(base_dir), path)):
            if os.path.exists(cand):
                t = A.ImageTexture.from_path(cand)
                t.path = path
                return t
        raise FileNotFoundError(path)
    raise ValueError(f"unknown texture tag {k!r}")


def _material(d, base_dir):
    k = d["material"]
    if k == "LambertianMat":
        return A.LambertianMat(_texture(d["albedo"], base_dir))
    if k == "MetalMat":
        return A.MetalMat(_v3(d["albedo"]), float(d["roughness"]))
    if k == "DielectricMat":
        return A.DielectricMat(float(d["ref_idx"]))
    if k == "EmissiveMat":
        return A.EmissiveMat(_texture(d["albedo"], base_dir))
    if k == "IsotropicMat":
        return A.IsotropicMat(_texture(d["texture"], base_dir))
    raise ValueError(f"unknown material tag {k!r}")


_RECTS = {"XYRect": A.XYRect, "XZRect": A.XZRect, "YZRect": A.YZRect}


def _rect(cls, d):
    r = cls(float(d["min"]["x"]), float(d["max"]["x"]), float(d["min"]["y"]), float(d["max"]["y"]), float(d["k"]),
            int(d["material"]))
    r.flip_normal = bool(d.get("flip_normal", False))
    return r


def _shape(d):
    k = d["object_type"]
    if k == "Sphere":
        return A.Sphere(float(d["radius"]), int(d["material"]))
    if k in _RECTS:
        return _rect(_RECTS[k], d)
    if k == "Rect3d":
        faces = d.get("faces") or []
        mat = None
        for f in faces:                                   # `faces` is derived data (rect3d.rs:18-80); take the material from it
            (_, rd), = f.items()
            mat = int(rd["material"])
            break
        if mat is None:
            raise ValueError("Rect3d without faces: no material")
        return A.Rect3d(np.asarray(_v3(d["pos"]), np.float32), np.asarray(_v3(d["size"]), np.float32), mat)
    if k == "TriangleMesh":
        verts = np.array([_v3(v) for v in d["verts"]], np.float32)
        normals = None if d.get("normals") is None else np.array([_v3(v) for v in d["normals"]], np.float32)
        uvs = None if d.get("uvs") is None else np.array([(float(v["x"]), float(v["y"])) for v in d["uvs"]], np.float32)
        return A.TriangleMesh(verts, np.array(d["indicies"], np.uint32), normals, uvs, int(d["material"]))
    if k == "ConstantMedium":
        return A.ConstantMedium(_shape(d["obj"]), float(d["density"]), int(d["material"]))
    if k == "Cone":
        return A.Cone(float(d["radius"]), float(d["height"]), int(d["material"]))
    if k == "Cylinder":
        return A.Cylinder(float(d["radius"]), float(d["height"]), int(d["material"]), float(d["max_phi"]))
    if k == "Disk":
        return A.Disk(float(d["radius"]), int(d["material"]), float(d["phi_max"]), float(d["inner_radius"]))
    raise ValueError(f"unknown object_type {k!r}")


def _environment(d):
    k = d["environment"]
    if k == "ColorEnv":
        return A.ColorEnv(_v3(d["color"]))
    if k == "SkyEnv":
        return A.SkyEnv(_v3(d["zenith_color"]), _v3(d["horizon_color"]))
    raise ValueError(f"unknown environment tag {k!r} (user-defined environments cannot cross to the GPU)")


def scene_from_dict(y, base_dir="."):
    scene = A.Scene()
    shapes = [_shape(ro["obj"]) for ro in y.get("render_objects") or []]     # unsupported shapes surface first
    for m in y.get("materials") or []:
        scene.add_material(_material(m, base_dir))
    for ro, shape in zip(y.get("render_objects") or [], shapes):
        o = A.RenderObject(shape)
        o.position_vec(_v3(ro["position"]))
        r = ro["rotation"]
        o.rotate(A.Rotor3(float(r["s"]), float(r["bv"]["xy"]), float(r["bv"]["xz"]), float(r["bv"]["yz"])))
        if ro.get("flip_normals"):
            o.flip_normals()
        scene.add_object(o)
    scene.set_environment(_environment(y["environment"]))
    return scene


def load_scene(path):
    """`serde_yaml::from_reader(file)` of src/main.rs:24-25."""
    with open(path) as f:
        loader = getattr(yaml, "CSafeLoader", yaml.SafeLoader)
        y = yaml.load(f, Loader=loader)
    return scene_from_dict(y, os.path.dirname(os.path.abspath(path)))


# ------------------------------------------------------------------------------------------- save
def _texture_d(t):
    if isinstance(t, A.ConstantTexture):
        return {"texture": "ConstantTexture", "color": _v3d(t.color)}
    if isinstance(t, A.CheckerTexture):
        return {"texture": "CheckerTexture", "odd": _texture_d(t.odd), "even": _texture_d(t.even), "scale": float(t.scale)}
    if isinstance(t, A.PerlinNoiseTexture):
        return {"texture": "PerlinNoiseTexture", "scale": float(t.scale)}
    if isinstance(t, A.TurbulenceTexture):
        return {"texture": "TurbulenceTexture", "depth": int(t.depth), "scale": float(t.scale)}
    if isinstance(t, A.MarbleTexture):
        return {"texture": "MarbleTexture", "depth": int(t.depth), "scale": float(t.scale)}
    if isinstance(t, A.ImageTexture):
        if t.path is None:
            raise ValueError("ImageTexture.path not specified")       # texture.rs:265 panics the same way
        return {"texture": "ImageTexture", "value": t.path}
    raise TypeError(type(t))


def _material_d(m):
    if isinstance(m, A.LambertianMat):
        return {"material": "LambertianMat", "albedo": _texture_d(m.albedo)}
    if isinstance(m, A.MetalMat):
        return {"material": "MetalMat", "albedo": _v3d(m.albedo), "roughness": float(m.roughness)}
    if isinstance(m, A.DielectricMat):
        return {"material": "DielectricMat", "ref_idx": float(m.ref_idx)}
    if isinstance(m, A.EmissiveMat):
        return {"material": "EmissiveMat", "albedo": _texture_d(m.albedo)}
    if isinstance(m, A.IsotropicMat):
        return {"material": "IsotropicMat", "texture": _texture_d(m.texture)}
    raise TypeError(type(m))


def _rect_d(r):
    return {"min": {"x": float(r.a_min), "y": float(r.b_min)}, "max": {"x": float(r.a_max), "y": float(r.b_max)},
            "k": float(r.k), "flip_normal": bool(r.flip_normal), "material": int(r.material)}


def _shape_d(s):
    if isinstance(s, A.Sphere):
        return {"object_type": "Sphere", "radius": float(s.radius), "material": int(s.material)}
    if isinstance(s, A.Cone):
        return {"object_type": "Cone", "radius": float(s.radius), "height": float(s.height), "material": int(s.material)}
    if isinstance(s, A.Cylinder):
        return {"object_type": "Cylinder", "radius": float(s.radius), "height": float(s.height), "max_phi": float(s.max_phi),
                "material": int(s.material)}
    if isinstance(s, A.Disk):
        return {"object_type": "Disk", "radius": float(s.radius), "phi_max": float(s.phi_max),
                "inner_radius": float(s.inner_radius), "material": int(s.material)}
    for name, cls in _RECTS.items():
        if isinstance(s, cls):
            return {"object_type": name, **_rect_d(s)}
    if isinstance(s, A.Rect3d):
        p, z, m = [float(x) for x in s.pos], [float(x) for x in s.size], int(s.material)
        def face(tag, a0, a1, b0, b1, k, flip):
            return {tag: {"min": {"x": a0, "y": b0}, "max": {"x": a1, "y": b1}, "k": k, "flip_normal": flip, "material": m}}
        faces = [face("XY", p[0], p[0] + z[0], p[1], p[1] + z[1], p[2] + z[2], False),      # rect3d.rs:19-77
                 face("XY", p[0], p[0] + z[0], p[1], p[1] + z[1], p[2], True),
                 face("XZ", p[0], p[0] + z[0], p[2], p[2] + z[2], p[1] + z[1], False),
                 face("XZ", p[0], p[0] + z[0], p[2], p[2] + z[2], p[1], True),
                 face("YZ", p[1], p[1] + z[1], p[2], p[2] + z[2], p[0] + z[0], False),
                 face("YZ", p[1], p[1] + z[1], p[2], p[2] + z[2], p[0], True)]
        return {"object_type": "Rect3d", "pos": _v3d(p), "size": _v3d(z), "faces": faces}
    if isinstance(s, A.TriangleMesh):
        return {"object_type": "TriangleMesh", "indicies": [int(i) for i in s.indicies],
                "verts": [_v3d(v) for v in s.verts],
                "normals": None if s.normals is None else [_v3d(v) for v in s.normals],
                "uvs": None if s.uvs is None else [{"x": float(v[0]), "y": float(v[1])} for v in s.uvs],
                "material": int(s.material)}
    if isinstance(s, A.ConstantMedium):
        return {"object_type": "ConstantMedium", "obj": _shape_d(s.obj), "density": float(s.density), "material": int(s.material)}
    raise TypeError(type(s))


def scene_to_dict(scene):
    e = scene.environment
    if isinstance(e, A.ColorEnv):
        env = {"environment": "ColorEnv", "color": _v3d(e.color)}
    elif isinstance(e, A.SkyEnv):
        env = {"environment": "SkyEnv", "zenith_color": _v3d(e.zenith_color), "horizon_color": _v3d(e.horizon_color)}
    else:
        raise TypeError("only ColorEnv / SkyEnv have a YAML form in the reference")
    return {
        "render_objects": [{"obj": _shape_d(ro.obj), "position": _v3d(ro._position),
                            "rotation": {"s": float(ro.rotation.s), "bv": {"xy": float(ro.rotation.xy), "xz": float(ro.rotation.xz),
                                                                           "yz": float(ro.rotation.yz)}},
                            "flip_normals": bool(ro._flip_normals)} for ro in scene.render_objects],
        "materials": [_material_d(m) for m in scene.materials],
        "environment": env,
    }


def save_scene(scene, path):
    with open(path, "w") as f:
        yaml.safe_dump(scene_to_dict(scene), f, sort_keys=False, default_flow_style=False)
