"""Host-side mirror of the reference's scene/renderer API for the render path.

Same names, argument order and error behaviour as the Rust builders so that code written
against ritobanrc/firework reads the same here:

    reference                                         this module
    ------------------------------------------------  ------------------------------------
    Scene::new / add_material / add_object /          Scene (src/scene.rs:19-91)
      add_volume / set_environment
    RenderObject::new(..).position().rotate()         RenderObject (src/scene.rs:270-334)
      .flip_normals()
    Sphere / XYRect / XZRect / YZRect / Rect3d /      src/objects/*.rs
      TriangleMesh / ConstantMedium
    LambertianMat / MetalMat / DielectricMat /        src/material.rs
      EmissiveMat / IsotropicMat
    ConstantTexture / CheckerTexture / Perlin.. /     src/texture.rs
      Turbulence.. / Marble.. / ImageTexture
    ColorEnv / SkyEnv / HdrEnvironment                src/environment.rs, examples/hdri_test.rs:22-82
    CameraSettings                                    src/camera.rs:18-71
    Renderer::default().width()..render(scene)        src/render.rs:59-218

`Renderer.render` flattens the scene into the C ABI of include/firework_hip.h and calls the HIP
library; there is no CPU fallback (a missing library or GPU raises).
"""
from __future__ import annotations

import ctypes as C
import math
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _abi as A

F32 = np.float32


def _v3(v) -> np.ndarray:
    a = np.asarray(v, dtype=F32).reshape(3)
    return a


# --------------------------------------------------------------------------- Rotor3 (ultraviolet)
@dataclass
class Rotor3:
    """ultraviolet::Rotor3 as serialised by src/serde_compat.rs:6-20 (`{s, bv: {xy, xz, yz}}`).

    Pinned by scenes/*.yml: from_rotation_P(t) = {s: cos(t/2), P: -sin(t/2)};
    from_euler_angles(roll, pitch, yaw) = R_xz(yaw) * R_yz(pitch) * R_xy(roll) (geometric product).
    Angles are radians; the reference's examples pass e.g. `-30.` (radians, an example bug kept as written).
    """
    s: float = 1.0
    xy: float = 0.0
    xz: float = 0.0
    yz: float = 0.0

    @staticmethod
    def identity() -> "Rotor3":
        return Rotor3()

    @staticmethod
    def _plane(angle, which) -> "Rotor3":
        # ultraviolet: Rotor3::new(cos(a/2), unit_plane * -sin(a/2)); the products with the two zero
        # plane components keep their sign (teapot.yml holds `xy: -0.0` for from_rotation_xz(90.)).
        half = F32(angle) / F32(2.0)
        ms, c = -F32(math.sin(float(half))), F32(math.cos(float(half)))
        comps = {k: float(F32(1.0 if k == which else 0.0) * ms) for k in ("xy", "xz", "yz")}
        return Rotor3(float(c), comps["xy"], comps["xz"], comps["yz"])

    @staticmethod
    def from_rotation_xy(angle) -> "Rotor3":
        return Rotor3._plane(angle, "xy")

    @staticmethod
    def from_rotation_xz(angle) -> "Rotor3":
        return Rotor3._plane(angle, "xz")

    @staticmethod
    def from_rotation_yz(angle) -> "Rotor3":
        return Rotor3._plane(angle, "yz")

    def __mul__(self, q: "Rotor3") -> "Rotor3":
        """Geometric product of two rotors on the basis (1, e12, e13, e23), in f32 with every sum a fused
        multiply-add chain (ultraviolet builds its products from `mul_add`): the association that reproduces the
        rotor serialised in the reference's scenes/conics.yml BIT FOR BIT (plain f32 sums in any order miss `s` and
        `xz` by one ulp; tests/test_oracle_known_answers.py::test_rotor_constructors_match_reference_yaml)."""
        a, b = self, q

        def fma32(x, y, c):
            """f32 fma(x, y, c) with ONE rounding (std::fma in include/firework.hpp): the product of two f32 is exact in f64, the
            f64 sum is made round-to-odd with the error term of TwoSum, and an odd-rounded f64 rounds to the correct f32."""
            p = float(x) * float(y)
            s_ = p + float(c)
            bb = s_ - p
            err = (p - (s_ - bb)) + (float(c) - bb)
            if err != 0.0 and math.isfinite(s_):
                m = struct.unpack("<q", struct.pack("<d", s_))[0]
                if (m & 1) == 0:                                   # even mantissa and inexact: step to the odd neighbour on the error's side
                    s_ = math.nextafter(s_, math.inf if err > 0 else -math.inf)
            return F32(s_)

        def chain(t0, t1, t2, t3):          # fma(t0, fma(t3, fma(t2, round(t1)))): one rounding per step
            acc = F32(t1[0] * t1[1])
            for x, y in (t2, t3, t0):
                acc = fma32(x, y, acc)
            return float(acc)
        f = lambda x: float(F32(x))
        a_s, a_xy, a_xz, a_yz, b_s, b_xy, b_xz, b_yz = (f(x) for x in (a.s, a.xy, a.xz, a.yz, b.s, b.xy, b.xz, b.yz))
        return Rotor3(
            s=chain((a_s, b_s), (-a_xy, b_xy), (-a_xz, b_xz), (-a_yz, b_yz)),
            xy=chain((a_xy, b_s), (a_s, b_xy), (a_yz, b_xz), (-a_xz, b_yz)),
            xz=chain((a_xz, b_s), (a_s, b_xz), (-a_yz, b_xy), (a_xy, b_yz)),
            yz=chain((a_yz, b_s), (a_s, b_yz), (a_xz, b_xy), (-a_xy, b_xz)),
        )

    @staticmethod
    def from_euler_angles(roll, pitch, yaw) -> "Rotor3":
        return Rotor3.from_rotation_xz(yaw) * Rotor3.from_rotation_yz(pitch) * Rotor3.from_rotation_xy(roll)

    def reversed(self) -> "Rotor3":
        return Rotor3(self.s, -self.xy, -self.xz, -self.yz)

    def to_abi(self) -> A.fw_rotor3:
        return A.fw_rotor3(self.s, self.xy, self.xz, self.yz)


# --------------------------------------------------------------------------- textures
class Texture:
    pass


@dataclass
class ConstantTexture(Texture):
    color: np.ndarray

    def __init__(self, color):
        self.color = _v3(color)

    @staticmethod
    def new(color):
        return ConstantTexture(color)

    @staticmethod
    def from_rgb(r, g, b):
        return ConstantTexture((r, g, b))


@dataclass
class CheckerTexture(Texture):
    odd: Texture
    even: Texture
    scale: float

    @staticmethod
    def new(odd, even, scale):
        return CheckerTexture(odd, even, scale)

    @staticmethod
    def with_colors(odd, even, scale):
        return CheckerTexture(ConstantTexture(odd), ConstantTexture(even), scale)


@dataclass
class PerlinNoiseTexture(Texture):
    scale: float

    @staticmethod
    def new(scale):
        return PerlinNoiseTexture(scale)


@dataclass
class TurbulenceTexture(Texture):
    depth: int
    scale: float

    @staticmethod
    def new(depth, scale):
        return TurbulenceTexture(depth, scale)


@dataclass
class MarbleTexture(Texture):
    depth: int
    scale: float

    @staticmethod
    def new(depth, scale):
        return MarbleTexture(depth, scale)


class ImageTexture(Texture):
    """src/texture.rs:270-310.  `image`: HxWx3 uint8 array, row 0 = top."""

    def __init__(self, image, path: Optional[str] = None):
        img = np.ascontiguousarray(np.asarray(image, dtype=np.uint8))
        if img.ndim != 3 or img.shape[2] < 3:
            raise ValueError("ImageTexture expects an HxWx3 uint8 array")
        self.image = np.ascontiguousarray(img[:, :, :3])
        self.path = path

    @staticmethod
    def new(image):
        return ImageTexture(image)

    @staticmethod
    def from_path(path):
        from PIL import Image  # host-side decode only (the reference uses the `image` crate)
        with Image.open(path) as im:
            return ImageTexture(np.asarray(im.convert("RGB")), path=str(path))


# --------------------------------------------------------------------------- materials
class Material:
    pass


@dataclass
class LambertianMat(Material):
    albedo: Texture

    @staticmethod
    def new(albedo: Texture):
        return LambertianMat(albedo)

    @staticmethod
    def with_color(albedo):
        return LambertianMat(ConstantTexture(albedo))


@dataclass
class MetalMat(Material):
    albedo: np.ndarray
    roughness: float

    def __init__(self, albedo, roughness):
        self.albedo = _v3(albedo)
        self.roughness = float(roughness)

    @staticmethod
    def new(albedo, roughness):
        return MetalMat(albedo, roughness)


@dataclass
class DielectricMat(Material):
    ref_idx: float

    @staticmethod
    def new(ref_idx):
        return DielectricMat(ref_idx)


@dataclass
class EmissiveMat(Material):
    albedo: Texture

    @staticmethod
    def new(albedo: Texture):
        return EmissiveMat(albedo)

    @staticmethod
    def with_color(albedo):
        return EmissiveMat(ConstantTexture(albedo))


@dataclass
class IsotropicMat(Material):
    texture: Texture

    @staticmethod
    def new(texture: Texture):
        return IsotropicMat(texture)


# --------------------------------------------------------------------------- shapes
class Shape:
    pass


@dataclass
class Sphere(Shape):
    radius: float
    material: int

    @staticmethod
    def new(radius, material):
        return Sphere(radius, material)


_RADS_PER_DEG = F32(F32(math.pi) / F32(180.0))     # f32::to_radians: self * (PI / 180)


def to_radians(deg) -> float:
    """Rust `f32::to_radians` (the examples write `18_f32.to_radians()`): one f32 multiply by PI/180."""
    return float(F32(deg) * _RADS_PER_DEG)


@dataclass
class Cone(Shape):
    """src/objects/cone.rs:9-25"""
    radius: float
    height: float
    material: int

    @staticmethod
    def new(radius, height, material):
        return Cone(radius, height, material)


@dataclass
class Cylinder(Shape):
    """src/objects/cylinder.rs:10-38 (max_phi in radians)"""
    radius: float
    height: float
    material: int
    max_phi: float = float(F32(360.0) * _RADS_PER_DEG)

    @staticmethod
    def new(radius, height, material):
        return Cylinder(radius, height, material)

    @staticmethod
    def partial(radius, height, phi, material):
        return Cylinder(radius, height, material, float(F32(phi) * _RADS_PER_DEG))


@dataclass
class Disk(Shape):
    """src/objects/disk.rs:10-37 (phi_max in radians)"""
    radius: float
    material: int
    phi_max: float = float(F32(2.0) * F32(math.pi))
    inner_radius: float = 0.0

    @staticmethod
    def new(radius, material):
        return Disk(radius, material)

    @staticmethod
    def partial(radius, phi, inner_radius, material):
        return Disk(radius, material, float(F32(phi) * _RADS_PER_DEG), float(inner_radius))


@dataclass
class _AARect(Shape):
    """AARect<A1,A2>::new(a1_min, a1_max, a2_min, a2_max, k, material) (src/objects/rect.rs:24-39)."""
    a_min: float
    a_max: float
    b_min: float
    b_max: float
    k: float
    material: int
    flip_normal: bool = False
    KIND = -1

    @classmethod
    def new(cls, a_min, a_max, b_min, b_max, k, material):
        return cls(a_min, a_max, b_min, b_max, k, material)


class XYRect(_AARect):
    KIND = A.FW_SHAPE_XYRECT


class XZRect(_AARect):
    KIND = A.FW_SHAPE_XZRECT


class YZRect(_AARect):
    KIND = A.FW_SHAPE_YZRECT


@dataclass
class Rect3d(Shape):
    pos: np.ndarray
    size: np.ndarray
    material: int

    @staticmethod
    def with_size(size, material):
        """src/objects/rect3d.rs:83-85"""
        return Rect3d(np.zeros(3, F32), _v3(size), material)


class TriangleMesh(Shape):
    """src/objects/mesh.rs:12-72"""

    def __init__(self, verts, indicies, normals=None, uvs=None, material=0):
        self.verts = np.ascontiguousarray(np.asarray(verts, dtype=F32).reshape(-1, 3))
        self.indicies = np.ascontiguousarray(np.asarray(indicies, dtype=np.uint32).reshape(-1))
        n = self.verts.shape[0]
        self.normals = None if normals is None else np.ascontiguousarray(np.asarray(normals, dtype=F32).reshape(-1, 3))
        self.uvs = None if uvs is None else np.ascontiguousarray(np.asarray(uvs, dtype=F32).reshape(-1, 2))
        if self.normals is not None and self.normals.shape[0] != n:
            raise ValueError("TriangleMesh::new() -- normals.len() must equal verts.len()")
        if self.uvs is not None and self.uvs.shape[0] != n:
            raise ValueError("TriangleMesh::new() -- uvs.len() must equal verts.len()")
        self.material = int(material)

    @staticmethod
    def new(verts, indicies, normals, uvs, material):
        return TriangleMesh(verts, indicies, normals, uvs, material)

    def translate(self, pos):
        self.verts = self.verts + _v3(pos)[None, :]
        return self

    def num_verts(self):
        return int(self.verts.shape[0])

    def num_tris(self):
        return int(self.indicies.shape[0] // 3)


@dataclass
class ConstantMedium(Shape):
    """src/objects/volume.rs:10-41; built by Scene.add_volume (src/scene.rs:47-62)."""
    obj: Shape
    density: float
    material: int


# --------------------------------------------------------------------------- environments
class Environment:
    pass


@dataclass
class ColorEnv(Environment):
    color: np.ndarray = field(default_factory=lambda: np.zeros(3, F32))

    def __init__(self, color=(0.0, 0.0, 0.0)):
        self.color = _v3(color)

    @staticmethod
    def new(color):
        return ColorEnv(color)


@dataclass
class SkyEnv(Environment):
    zenith_color: np.ndarray
    horizon_color: np.ndarray

    def __init__(self, zenith_color=(0.5, 0.7, 1.0), horizon_color=(1.0, 1.0, 1.0)):
        self.zenith_color = _v3(zenith_color)
        self.horizon_color = _v3(horizon_color)

    @staticmethod
    def new(zenith_color, horizon_color):
        return SkyEnv(zenith_color, horizon_color)

    @staticmethod
    def default():
        return SkyEnv()


class HdrEnvironment(Environment):
    """examples/hdri_test.rs:22-82 (equirect, nearest lookup).  `pixels`: HxWx3 float32, row 0 = top."""

    def __init__(self, pixels):
        p = np.ascontiguousarray(np.asarray(pixels, dtype=F32))
        if p.ndim != 3 or p.shape[2] != 3:
            raise ValueError("HdrEnvironment expects an HxWx3 float32 array")
        self.pixels = p


# --------------------------------------------------------------------------- RenderObject / Scene
class RenderObject:
    """src/scene.rs:270-334"""

    def __init__(self, obj: Shape):
        self.obj = obj
        self._position = np.zeros(3, F32)
        self.rotation = Rotor3.identity()
        self._flip_normals = False

    @staticmethod
    def new(obj: Shape):
        return RenderObject(obj)

    def position(self, x, y, z):
        self._position = _v3((x, y, z))
        return self

    def position_vec(self, pos):
        self._position = _v3(pos)
        return self

    def rotate(self, rotor: Rotor3):
        self.rotation = rotor
        return self

    def flip_normals(self):
        self._flip_normals = not self._flip_normals
        return self


class Scene:
    """src/scene.rs:19-91"""

    def __init__(self):
        self.render_objects: List[RenderObject] = []
        self.materials: List[Material] = []
        self.environment: Environment = ColorEnv()  # Scene::new(): black ColorEnv (scene.rs:36)

    @staticmethod
    def new():
        return Scene()

    def add_object(self, obj: RenderObject) -> int:
        self.render_objects.append(obj)
        return len(self.render_objects) - 1

    def add_volume(self, obj: RenderObject, density, texture: Texture) -> int:
        mat = self.add_material(IsotropicMat(texture))
        obj.obj = ConstantMedium(obj.obj, float(density), mat)
        return self.add_object(obj)

    def get_object(self, idx):
        return self.render_objects[idx]

    def add_material(self, mat: Material) -> int:
        self.materials.append(mat)
        return len(self.materials) - 1

    def get_material(self, idx):
        return self.materials[idx]

    def set_environment(self, env: Environment):
        self.environment = env

    # ---- flatten to the C ABI -------------------------------------------------
    def to_desc(self) -> "SceneDesc":
        return SceneDesc(self)


class SceneDesc:
    """Owns the ctypes arrays (and the numpy buffers they point into) of one fw_scene_desc."""

    def __init__(self, scene: Scene):
        self._keep = []
        texs: List[A.fw_texture] = []

        def add_tex(t: Texture) -> int:
            ft = A.fw_texture()
            if isinstance(t, ConstantTexture):
                ft.kind = A.FW_TEX_CONSTANT
                ft.color = A.vec3(t.color)
            elif isinstance(t, CheckerTexture):
                ft.kind = A.FW_TEX_CHECKER
                ft.scale = t.scale
                ft.odd = add_tex(t.odd)
                ft.even = add_tex(t.even)
            elif isinstance(t, PerlinNoiseTexture):
                ft.kind = A.FW_TEX_PERLIN
                ft.scale = t.scale
            elif isinstance(t, TurbulenceTexture):
                ft.kind = A.FW_TEX_TURBULENCE
                ft.scale = t.scale
                ft.depth = t.depth
            elif isinstance(t, MarbleTexture):
                ft.kind = A.FW_TEX_MARBLE
                ft.scale = t.scale
                ft.depth = t.depth
            elif isinstance(t, ImageTexture):
                ft.kind = A.FW_TEX_IMAGE
                ft.img_h, ft.img_w = t.image.shape[0], t.image.shape[1]
                ft.img_rgb8 = t.image.ctypes.data_as(C.POINTER(C.c_uint8))
                self._keep.append(t.image)
            else:
                raise TypeError(f"unknown texture {type(t).__name__}")
            texs.append(ft)
            return len(texs) - 1

        mats = []
        for m in scene.materials:
            fm = A.fw_material()
            fm.texture = -1
            if isinstance(m, LambertianMat):
                fm.kind = A.FW_MAT_LAMBERTIAN
                fm.texture = add_tex(m.albedo)
            elif isinstance(m, MetalMat):
                fm.kind = A.FW_MAT_METAL
                fm.albedo = A.vec3(m.albedo)
                fm.roughness = m.roughness
            elif isinstance(m, DielectricMat):
                fm.kind = A.FW_MAT_DIELECTRIC
                fm.ref_idx = m.ref_idx
            elif isinstance(m, EmissiveMat):
                fm.kind = A.FW_MAT_EMISSIVE
                fm.texture = add_tex(m.albedo)
            elif isinstance(m, IsotropicMat):
                fm.kind = A.FW_MAT_ISOTROPIC
                fm.texture = add_tex(m.texture)
            else:
                raise TypeError(f"unknown material {type(m).__name__}")
            mats.append(fm)

        shapes: List[A.fw_shape] = []

        shape_index = {}      # one fw_shape per Python shape object: objects that share a TriangleMesh share its BLAS

        def add_shape(s: Shape) -> int:
            if id(s) in shape_index:
                return shape_index[id(s)]
            fs = A.fw_shape()
            fs.inner = -1
            if isinstance(s, Sphere):
                fs.kind, fs.radius, fs.material = A.FW_SHAPE_SPHERE, s.radius, s.material
            elif isinstance(s, Cone):
                fs.kind, fs.material, fs.radius, fs.height = A.FW_SHAPE_CONE, s.material, s.radius, s.height
            elif isinstance(s, Cylinder):
                fs.kind, fs.material, fs.radius, fs.height, fs.phi_max = A.FW_SHAPE_CYLINDER, s.material, s.radius, s.height, s.max_phi
            elif isinstance(s, Disk):
                fs.kind, fs.material, fs.radius, fs.phi_max, fs.inner_radius = A.FW_SHAPE_DISK, s.material, s.radius, s.phi_max, s.inner_radius
            elif isinstance(s, _AARect):
                fs.kind, fs.material = s.KIND, s.material
                fs.a_min, fs.a_max, fs.b_min, fs.b_max, fs.k = s.a_min, s.a_max, s.b_min, s.b_max, s.k
                fs.flip_normal = int(bool(s.flip_normal))
            elif isinstance(s, Rect3d):
                fs.kind, fs.material = A.FW_SHAPE_RECT3D, s.material
                fs.pos, fs.size = A.vec3(s.pos), A.vec3(s.size)
            elif isinstance(s, TriangleMesh):
                fs.kind, fs.material = A.FW_SHAPE_TRIANGLE_MESH, s.material
                fs.verts = s.verts.ctypes.data_as(C.POINTER(C.c_float))
                fs.n_verts = s.verts.shape[0]
                fs.indices = s.indicies.ctypes.data_as(C.POINTER(C.c_uint32))
                fs.n_indices = s.indicies.shape[0]
                self._keep += [s.verts, s.indicies]
                if s.normals is not None:
                    fs.normals = s.normals.ctypes.data_as(C.POINTER(C.c_float))
                    self._keep.append(s.normals)
                if s.uvs is not None:
                    fs.uvs = s.uvs.ctypes.data_as(C.POINTER(C.c_float))
                    self._keep.append(s.uvs)
            elif isinstance(s, ConstantMedium):
                fs.kind, fs.material, fs.density = A.FW_SHAPE_CONSTANT_MEDIUM, s.material, s.density
                fs.inner = add_shape(s.obj)
            else:
                raise TypeError(f"unknown shape {type(s).__name__}")
            shapes.append(fs)
            shape_index[id(s)] = len(shapes) - 1
            return len(shapes) - 1

        objs = []
        for ro in scene.render_objects:
            fo = A.fw_object()
            fo.shape = add_shape(ro.obj)
            fo.position = A.vec3(ro._position)
            fo.rotation = ro.rotation.to_abi()
            fo.flip_normals = int(ro._flip_normals)
            objs.append(fo)

        env = A.fw_environment()
        e = scene.environment
        if isinstance(e, ColorEnv):
            env.kind, env.color = A.FW_ENV_COLOR, A.vec3(e.color)
        elif isinstance(e, SkyEnv):
            env.kind, env.zenith, env.horizon = A.FW_ENV_SKY, A.vec3(e.zenith_color), A.vec3(e.horizon_color)
        elif isinstance(e, HdrEnvironment):
            env.kind = A.FW_ENV_HDR
            env.hdr_h, env.hdr_w = e.pixels.shape[0], e.pixels.shape[1]
            env.hdr_rgb = e.pixels.ctypes.data_as(C.POINTER(C.c_float))
            self._keep.append(e.pixels)
        else:
            raise TypeError(f"unknown environment {type(e).__name__}")

        def arr(ctype, items):
            a = (ctype * max(1, len(items)))(*items)
            self._keep.append(a)
            return a

        self.objects, self.shapes = arr(A.fw_object, objs), arr(A.fw_shape, shapes)
        self.materials, self.textures = arr(A.fw_material, mats), arr(A.fw_texture, texs)
        d = A.fw_scene_desc()
        d.objects, d.n_objects = self.objects, len(objs)
        d.shapes, d.n_shapes = self.shapes, len(shapes)
        d.materials, d.n_materials = self.materials, len(mats)
        d.textures, d.n_textures = self.textures, len(texs)
        d.environment = env
        self.desc = d

    def ptr(self):
        return C.byref(self.desc)

    def content_hash(self) -> str:
        """sha256 over everything a render depends on: every struct field that is not a pointer, and the arrays the
        pointers name (vertices, indices, normals, uvs, image and HDR pixels).  Identifies a scene in a checkpoint."""
        import hashlib
        h = hashlib.sha256()

        def feed(obj):
            for name, _ in obj._fields_:
                v = getattr(obj, name)
                if isinstance(v, C.Structure):
                    feed(v)
                elif isinstance(v, C._Pointer):
                    if v:                                     # the array it names is hashed below: it must be one of the kept ones
                        addr = C.cast(v, C.c_void_p).value
                        assert any(isinstance(a, np.ndarray) and a.ctypes.data == addr for a in self._keep), f"{name}: pointer without a kept array"
                    continue
                elif isinstance(v, C.Array):
                    h.update(bytes(v))
                else:
                    h.update(repr(v).encode())
        d = self.desc
        for arr, n in ((self.objects, d.n_objects), (self.shapes, d.n_shapes), (self.materials, d.n_materials), (self.textures, d.n_textures)):
            for i in range(n):
                feed(arr[i])
        feed(d.environment)
        for a in self._keep:
            if isinstance(a, np.ndarray):
                h.update(str(a.shape).encode())
                h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()


# --------------------------------------------------------------------------- camera / renderer
class CameraSettings:
    """src/camera.rs:18-71"""

    def __init__(self):
        self._cam_pos = _v3((0.0, 0.0, -10.0))
        self._look_at = _v3((0.0, 0.0, 0.0))
        self._vfov = 30.0
        self._aperture = 0.0
        self._focus_dist = 10.0

    @staticmethod
    def default():
        return CameraSettings()

    def cam_pos(self, v):
        self._cam_pos = _v3(v)
        return self

    def look_at(self, v):
        self._look_at = _v3(v)
        return self

    def field_of_view(self, vfov):
        self._vfov = float(vfov)
        return self

    def aperture(self, a):
        self._aperture = float(a)
        return self

    def focus_dist(self, d):
        self._focus_dist = float(d)
        return self

    def to_abi(self) -> A.fw_camera_settings:
        return A.fw_camera_settings(A.vec3(self._cam_pos), A.vec3(self._look_at), self._vfov, self._aperture,
                                    self._focus_dist)


@dataclass
class RenderResult:
    rgb8: np.ndarray       # (N,3) uint8  — `Vec<Color>` (render.rs:109)
    gamma: np.ndarray      # (N,3) float32 post-gamma clamped (render.rs:185-187)
    linear: np.ndarray     # (N,3) float32 pre-gamma mean (render.rs:184)
    stats: dict
    width: int
    height: int

    def image(self) -> np.ndarray:
        return self.rgb8.reshape(self.height, self.width, 3)


class Renderer:
    """src/render.rs:59-218.  Builder methods carry the reference's names; read the current
    values from `.settings`."""

    def __init__(self):
        # Default (render.rs:199-218)
        self.settings = dict(width=1920, height=1080, samples=128, multithreaded=True, use_bvh=False, gamma=2.2,
                             seed=0, paths_per_batch=0, flags=0)
        self._camera = CameraSettings()

    @staticmethod
    def default():
        return Renderer()

    def width(self, w):
        self.settings["width"] = int(w)
        return self

    def height(self, h):
        self.settings["height"] = int(h)
        return self

    def samples(self, s):
        self.settings["samples"] = int(s)
        return self

    def multithreaded(self, m):
        self.settings["multithreaded"] = bool(m)
        return self

    def use_bvh(self, b):
        self.settings["use_bvh"] = bool(b)
        return self

    def gamma(self, g):
        self.settings["gamma"] = float(g)
        return self

    def camera(self, settings: CameraSettings):
        self._camera = settings
        return self

    # extensions of the GPU path (not in the reference)
    def seed(self, s):
        self.settings["seed"] = int(s)
        return self

    def paths_per_batch(self, n):
        self.settings["paths_per_batch"] = int(n)
        return self

    def time_kernels(self, on=True):
        self.settings["flags"] = (self.settings["flags"] | A.FW_FLAG_TIME_KERNELS) if on else (
            self.settings["flags"] & ~A.FW_FLAG_TIME_KERNELS)
        return self

    def count_deposits(self, on=True):
        """FW_FLAG_COUNT_DEPOSITS: fw_stats.deposits / bytes_shade become exact where zero deposits are elided (one extra pass)."""
        self.settings["flags"] = (self.settings["flags"] | A.FW_FLAG_COUNT_DEPOSITS) if on else (
            self.settings["flags"] & ~A.FW_FLAG_COUNT_DEPOSITS)
        return self

    def to_params(self, pixel_ids: Optional[np.ndarray] = None, rng_mode: int = A.FW_RNG_CTR) -> A.fw_render_params:
        s = self.settings
        p = A.fw_render_params()
        p.width, p.height, p.samples = s["width"], s["height"], s["samples"]
        p.gamma = s["gamma"]
        p.use_bvh = int(s["use_bvh"])
        p.multithreaded = int(s["multithreaded"])
        p.camera = self._camera.to_abi()
        p.seed = s["seed"]
        p.rng_mode = rng_mode
        p.paths_per_batch = s["paths_per_batch"]
        p.flags = s["flags"]
        if pixel_ids is not None:
            p.pixel_ids = pixel_ids.ctypes.data_as(C.POINTER(C.c_uint32))
            p.n_pixels = int(pixel_ids.shape[0])
        return p

    def render_full(self, scene, pixel_ids: Optional[Sequence[int]] = None, device: int = 0) -> RenderResult:
        """`Renderer::render` + the float buffers and counters the parity tests and bench need."""
        from . import _lib
        sd = scene if isinstance(scene, SceneDesc) else scene.to_desc()
        ids = None if pixel_ids is None else np.ascontiguousarray(np.asarray(pixel_ids, dtype=np.uint32))
        return _lib.render_scene(sd, self, ids, device)

    def render_progressive(self, scene, passes: int, device: int = 0, checkpoint: Optional[str] = None):
        """Progressive preview / resumable render (not in the reference): yields a RenderResult after each of
        `passes` passes that split the samples evenly; the last one is bit-identical to `render_full`.  With `checkpoint`
        (an .npz path) the accumulation buffer is saved after every pass and a matching file is resumed from."""
        import copy
        from . import _lib
        sd = scene if isinstance(scene, SceneDesc) else scene.to_desc()
        s = self.settings
        total, n = int(s["samples"]), int(s["width"]) * int(s["height"])
        passes = max(1, min(int(passes), total))
        accum, done = np.zeros((n, 4), np.float32), 0
        # what the accumulated sums depend on: image size, seed, the scene's content, the camera and use_bvh (the hit of a
        # ray that grazes a padded box can differ between the linear scan and the BVH, as in the reference).  gamma and the
        # total sample count do not enter the sums.
        import hashlib
        import os
        import sys
        cam = self._camera.to_abi()
        # ... and on the build: an accumulation buffer written by other kernels (another ABI, other kernel sources) must not be blended in
        tag = hashlib.sha256(repr((int(s["width"]), int(s["height"]), int(s["seed"]), bool(s["use_bvh"]), sd.content_hash(),
                                   bytes(cam), int(A.FW_ABI_VERSION), int(A.FW_RNG_CTR), _lib.build_id())).encode()).hexdigest()
        if checkpoint and not str(checkpoint).endswith(".npz"):
            checkpoint = str(checkpoint) + ".npz"                       # the name np.savez would write
        if checkpoint and os.path.exists(checkpoint):
            try:
                with np.load(checkpoint) as ck:
                    ok = ck["accum"].shape == accum.shape and str(ck["tag"]) == tag and 0 < int(ck["done"]) <= total
                    if ok:
                        accum, done = np.ascontiguousarray(ck["accum"], dtype=np.float32), int(ck["done"])
                    else:
                        print(f"checkpoint {checkpoint}: written for another scene, camera, size, seed or use_bvh "
                              f"(or holds more samples than asked for) - ignored, starting from sample 0", file=sys.stderr)
            except Exception as e:                                      # truncated or foreign file
                print(f"checkpoint {checkpoint}: unreadable ({type(e).__name__}: {e}) - ignored, starting from sample 0", file=sys.stderr)
        ds = _lib.DeviceScene(sd, device)
        try:
            bounds = [round(total * k / passes) for k in range(passes + 1)]
            for k in range(passes):
                lo, hi = max(bounds[k], done), bounds[k + 1]
                if hi <= lo:
                    continue
                r = copy.copy(self); r.settings = dict(self.settings); r.settings["samples"] = hi - lo
                res = ds.render_progressive(r, lo, accum)
                done = hi
                if checkpoint:                                          # atomic: a kill during the write leaves the old file
                    tmp = f"{checkpoint}.tmp.{os.getpid()}"
                    with open(tmp, "wb") as f:
                        np.savez(f, accum=accum, done=np.int64(done), tag=np.array(tag))
                    os.replace(tmp, checkpoint)
                yield res
        finally:
            ds.close()

    def render(self, scene, device: int = 0) -> np.ndarray:
        """`pub fn render(&self, scene: Scene) -> Vec<Color>` (render.rs:109): (W*H, 3) uint8, row 0 = top."""
        return self.render_full(scene, None, device).rgb8


def save_image(render: np.ndarray, path, width: int, height: int):
    """src/window.rs:59-66"""
    from PIL import Image
    Image.fromarray(np.asarray(render, dtype=np.uint8).reshape(height, width, 3), "RGB").save(path)
