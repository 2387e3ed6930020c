"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
firework_amd/.  Shares only the data format (firework_amd/_abi.py == include/firework_hip.h)."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from firework_amd import _abi as A  # noqa: E402

LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None
F = C.c_float
PF = C.POINTER(C.c_float)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make -C oracle`")
        lib = C.CDLL(LIB_PATH)
        lib.fwo_render.restype = C.c_int
        lib.fwo_render.argtypes = [C.POINTER(A.fw_scene_desc), C.POINTER(A.fw_render_params), C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.POINTER(A.fw_stats)]
        lib.fwo_strerror.restype = C.c_char_p
        lib.fwo_schlick.restype = F
        lib.fwo_schlick.argtypes = [F, F]
        lib.fwo_solve_quadratic.argtypes = [F, F, F, PF]
        lib.fwo_perlin_noise.restype = F
        lib.fwo_turb.restype = F
        lib.fwo_turb.argtypes = [C.c_uint32, PF]
        lib.fwo_color_to_u32.restype = C.c_uint32
        lib.fwo_refract.argtypes = [PF, PF, F, PF]
        lib.fwo_rand4.argtypes = [C.c_uint64] + [C.c_uint32] * 5 + [C.POINTER(C.c_uint32), PF]
        lib.fwo_lcg_stream.argtypes = [C.c_uint64, C.c_int, PF]
        lib.fwo_texture_sample.argtypes = [C.POINTER(A.fw_scene_desc), C.c_int32, F, F, PF, PF]
        lib.fwo_trace.argtypes = [C.POINTER(A.fw_scene_desc), C.c_int, C.c_uint64, C.c_uint32, PF, PF]
        _lib = lib
    return _lib


def _fa(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
    return a, a.ctypes.data_as(PF)


class OracleError(RuntimeError):
    def __init__(self, status):
        self.status = status
        super().__init__(f"oracle error {status}: {load().fwo_strerror(status).decode()}")


def native_timing_build():
    """A second build of the SAME source with -O3 -march=native, made on the host it runs on, for bench.py's cpu_baseline leg ONLY — how fast
    the restatement runs when the compiler may use the host's own instructions.  Never the checker: every comparison in tests/, smoke() and
    bench.py's parity leg uses liboracle.so (-O2 -ffp-contract=off, oracle/Makefile).  -> path of the library, or None if it cannot be built."""
    import hashlib
    import subprocess
    import tempfile
    src = os.path.join(_HERE, "fw_oracle.cpp")
    tag = hashlib.sha256(open(src, "rb").read()).hexdigest()[:12]
    out = os.path.join(tempfile.gettempdir(), f"liboracle_native_{tag}_{os.getuid()}.so")
    if not os.path.exists(out):
        cmd = ["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-shared", "-o", out + ".tmp", src]
        try:
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            os.replace(out + ".tmp", out)
        except Exception:
            return None
    return out


def render(scene, renderer, pixel_ids=None, rng_mode=A.FW_RNG_CTR, n_threads=0, timing_lib=None):
    """Renderer::render on the CPU oracle.  `scene`: firework_amd.api.Scene or SceneDesc.
    timing_lib: path of native_timing_build()'s library (bench.py's cpu_baseline leg only: timed, never compared)."""
    from firework_amd.api import RenderResult, SceneDesc
    lib = load()
    if timing_lib:
        lib = C.CDLL(timing_lib)
        lib.fwo_render.restype = C.c_int
        lib.fwo_render.argtypes = [C.POINTER(A.fw_scene_desc), C.POINTER(A.fw_render_params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(A.fw_stats)]
    sd = scene if isinstance(scene, SceneDesc) else scene.to_desc()
    ids = None if pixel_ids is None else np.ascontiguousarray(np.asarray(pixel_ids, dtype=np.uint32))
    p = renderer.to_params(ids, rng_mode)
    n = int(ids.shape[0]) if ids is not None else p.width * p.height
    rgb8 = np.empty((n, 3), np.uint8)
    gam = np.empty((n, 3), np.float32)
    lin = np.empty((n, 3), np.float32)
    st = A.fw_stats()
    rc = lib.fwo_render(sd.ptr(), C.byref(p), n_threads, rgb8.ctypes.data, gam.ctypes.data, lin.ctypes.data, C.byref(st))
    if rc != 0:
        raise OracleError(rc)
    return RenderResult(rgb8, gam, lin, st.as_dict(), p.width, p.height)


def coord_from_index(idx, w, h):
    out = (C.c_uint64 * 2)()
    load().fwo_coord_from_index(C.c_uint64(idx), C.c_uint64(w), C.c_uint64(h), out)
    return int(out[0]), int(out[1])


def color_from_vec3(c):
    a, p = _fa(c)
    out = (C.c_uint8 * 3)()
    load().fwo_color_from_vec3(p, out)
    return tuple(out)


def color_to_u32(c):
    return int(load().fwo_color_to_u32((C.c_uint8 * 3)(*c)))


def solve_quadratic(a, b, c):
    r = (F * 2)()
    n = load().fwo_solve_quadratic(a, b, c, r)
    return [float(r[i]) for i in range(n)]


def schlick(cosine, ref_idx):
    return float(load().fwo_schlick(cosine, ref_idx))


def reflect(v, n):
    (_, pv), (_, pn) = _fa(v), _fa(n)
    out = (F * 3)()
    load().fwo_reflect(pv, pn, out)
    return np.array(out, np.float32)


def refract(v, n, ni_over_nt):
    (a, pv), (b, pn) = _fa(v), _fa(n)
    out = (F * 3)()
    ok = load().fwo_refract(pv, pn, ni_over_nt, out)
    return np.array(out, np.float32) if ok else None


def sphere_uv(p):
    a, pp = _fa(p)
    out = (F * 2)()
    load().fwo_sphere_uv(pp, out)
    return float(out[0]), float(out[1])


def max_component_idx(v):
    a, p = _fa(v)
    return int(load().fwo_max_component_idx(p))


def perlin_noise(p):
    a, pp = _fa(p)
    return float(load().fwo_perlin_noise(pp))


def turb(depth, p):
    a, pp = _fa(p)
    return float(load().fwo_turb(depth, pp))


def rotor_into_matrix(rotor):
    """Returns the 3x3 matrix M (row-major numpy) with v' = M @ v."""
    r = rotor.to_abi()
    out = (F * 9)()
    load().fwo_rotor_into_matrix(C.byref(r), out)
    cols = np.array(out, np.float32).reshape(3, 3)
    return cols.T.copy()


def rand4(seed, pixel, sample, purpose, segment, index):
    u = (C.c_uint32 * 4)()
    f = (F * 4)()
    load().fwo_rand4(seed, pixel, sample, purpose, segment, index, u, f)
    return np.array(u, np.uint32), np.array(f, np.float32)


def lcg_stream(seed, n):
    out = np.empty(n, np.float32)
    load().fwo_lcg_stream(seed, n, out.ctypes.data_as(PF))
    return out


def camera(cam_settings, w, h):
    s = cam_settings.to_abi()
    out = (F * 22)()
    load().fwo_camera(C.byref(s), C.c_uint32(w), C.c_uint32(h), out)
    a = np.array(out, np.float32)
    names = ["position", "lower_left", "horizontal", "vertical", "u", "v", "w"]
    d = {n: a[3 * i:3 * i + 3] for i, n in enumerate(names)}
    d["lens_radius"] = float(a[21])
    return d


def env_sample(scene, direction):
    sd = scene.to_desc()
    a, p = _fa(direction)
    out = (F * 3)()
    rc = load().fwo_env_sample(sd.ptr(), p, out)
    if rc:
        raise OracleError(rc)
    return np.array(out, np.float32)


def texture_sample(scene_desc, tex_index, u, v, point):
    a, p = _fa(point)
    out = (F * 3)()
    rc = load().fwo_texture_sample(scene_desc.ptr(), tex_index, u, v, p, out)
    if rc:
        raise OracleError(rc)
    return np.array(out, np.float32)


def bvh_stats(scene):
    sd = scene.to_desc()
    out = (C.c_uint32 * 6)()
    rc = load().fwo_bvh_stats(sd.ptr(), out)
    if rc:
        raise OracleError(rc)
    return dict(zip(["nodes", "leaves", "double_leaves", "branches", "depth", "blas_nodes"], [int(x) for x in out]))


def mesh_bvh_stats(scene, obj):
    sd = scene.to_desc()
    out = (C.c_uint32 * 5)()
    rc = load().fwo_mesh_bvh_stats(sd.ptr(), C.c_uint32(obj), out)
    if rc:
        raise OracleError(rc)
    return dict(zip(["nodes", "leaves", "double_leaves", "branches", "depth"], [int(x) for x in out]))


def object_aabbs(scene):
    sd = scene.to_desc()
    n = sd.desc.n_objects
    out = np.empty((n, 6), np.float32)
    rc = load().fwo_object_aabbs(sd.ptr(), out.ctypes.data_as(PF))
    if rc:
        raise OracleError(rc)
    return out


def trace(scene, rays, use_bvh=False, seed=0):
    """One root.hit() per ray.  rays: (n,6) origin+dir.  Returns (n,10): hit,t,point,normal,material,u."""
    sd = scene if hasattr(scene, "ptr") else scene.to_desc()
    r = np.ascontiguousarray(np.asarray(rays, np.float32).reshape(-1, 6))
    out = np.empty((r.shape[0], 10), np.float32)
    rc = load().fwo_trace(sd.ptr(), int(use_bvh), seed, r.shape[0], r.ctypes.data_as(PF), out.ctypes.data_as(PF))
    if rc:
        raise OracleError(rc)
    return out


# ---- divergence probes (tools/diverge.py, tests/test_gpu_divergence.py) ----------------------------------------------
def _params(scene, renderer, pixel_ids=None):
    from firework_amd.api import SceneDesc
    sd = scene if isinstance(scene, SceneDesc) else scene.to_desc()
    ids = None if pixel_ids is None else np.ascontiguousarray(np.asarray(pixel_ids, dtype=np.uint32))
    return sd, ids, renderer.to_params(ids, A.FW_RNG_CTR)


def render_counts(scene, renderer, pixel_ids=None, n_threads=0):
    """Rays (root.hit calls) per pixel of a CTR render, pixel_ids order."""
    lib = load()
    sd, ids, p = _params(scene, renderer, pixel_ids)
    n = int(ids.shape[0]) if ids is not None else p.width * p.height
    out = np.zeros(n, np.uint32)
    rc = lib.fwo_render_counts(sd.ptr(), C.byref(p), C.c_int(n_threads), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise OracleError(rc)
    return out


def find_nan_paths(scene, renderer, pixel_ids=None, cap=4096, n_threads=0):
    """(pixel, sample) pairs of a CTR render whose path follows a ray with a NaN in it, and how many there are in all."""
    lib = load()
    sd, _ids, p = _params(scene, renderer, pixel_ids)
    out = np.zeros((cap, 2), np.uint32)
    n = C.c_uint32(0)
    rc = lib.fwo_find_nan_paths(sd.ptr(), C.byref(p), C.c_int(n_threads), out.ctypes.data_as(C.c_void_p), C.c_uint32(cap), C.byref(n))
    if rc:
        raise OracleError(rc)
    return out[: min(cap, n.value)], n.value


def path_lengths(scene, renderer, pixel, first_sample, n):
    """Segments (1..11) of the paths of samples first_sample .. first_sample+n-1 of one pixel."""
    lib = load()
    sd, _, p = _params(scene, renderer)
    out = np.zeros(n, np.uint8)
    rc = lib.fwo_path_lengths(sd.ptr(), C.byref(p), C.c_uint32(pixel), C.c_uint32(first_sample), C.c_uint32(n), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise OracleError(rc)
    return out


def trace_path(scene, renderer, pixel, sample):
    """Every segment of one path: (11, 16) float32 rows = ray o, d | hit flag, t, material | point | normal | reached; + colour."""
    lib = load()
    sd, _, p = _params(scene, renderer)
    out = np.zeros((11, 16), np.float32)
    col = np.zeros(3, np.float32)
    rc = lib.fwo_trace_path(sd.ptr(), C.byref(p), C.c_uint32(pixel), C.c_uint32(sample), out.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p))
    if rc:
        raise OracleError(rc)
    return out, col


LIBM_FN = dict(logf=0, log10f=1, sinf=2, asinf=3, acosf=4, atanf=5, atan2f=6, powf=7)


def libm(fn, x, y=None):
    """The host libm's float function `fn` over float32 arrays (what the reference's f32 methods call on this platform)."""
    lib = load()
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    yy = None if y is None else np.ascontiguousarray(y, np.float32)
    rc = lib.fwo_libm(C.c_int(LIBM_FN[fn]), C.c_uint32(x.size), x.ctypes.data_as(C.c_void_p), None if yy is None else yy.ctypes.data_as(C.c_void_p),
                      out.ctypes.data_as(C.c_void_p))
    if rc:
        raise OracleError(rc)
    return out
