"""Loads libfirework_hip.so (HIP kernels + C ABI) and wraps its entry points.

There is deliberately NO fallback: if the shared library is missing, fails to load, or no GPU is
visible, every call raises.  The CPU oracle under oracle/ is test infrastructure and is never
reachable from here.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FIREWORK_LIB") or os.path.join(_HERE, "lib", "libfirework_hip.so")
_lib = None


class FireworkError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"firework_hip error {status}: {detail}")


def load(preload=False, device=None):
    """Load the native library once.  torch is imported first so that the process holds exactly one
    HIP runtime (torch bundles libamdhip64.so.7; our library's DT_NEEDED resolves to that copy).
    Loading makes no HIP call (ABI v7).  preload=True also runs fw_init on `device` (default: LOCAL_RANK's, else 0) — context, code
    objects, kernel handles and the default path arena — which is what the CLI and bench.py want before their timed regions."""
    global _lib
    if _lib is not None:
        if preload:
            init(device)
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FireworkError(A.FW_ERR_NO_DEVICE, f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`"
                            " (or `make`) — there is no CPU fallback")
    if os.environ.get("FIREWORK_NO_TORCH", "0") != "1":
        import torch  # noqa: F401  (plumbing: one HIP runtime per process, streams, torch.distributed)
    lib = C.CDLL(LIB_PATH)
    lib.fw_abi_version.restype = C.c_int
    lib.fw_strerror.restype = C.c_char_p
    lib.fw_strerror.argtypes = [C.c_int]
    lib.fw_last_error.restype = C.c_char_p
    lib.fw_device_count.restype = C.c_int
    lib.fw_scene_create.restype = C.c_int
    lib.fw_scene_create.argtypes = [C.POINTER(A.fw_scene_desc), C.c_int, C.POINTER(C.c_void_p)]
    lib.fw_scene_destroy.restype = None
    lib.fw_scene_destroy.argtypes = [C.c_void_p]
    lib.fw_render.restype = C.c_int
    lib.fw_render.argtypes = [C.c_void_p, C.POINTER(A.fw_render_params), C.c_void_p, C.c_void_p, C.c_void_p,
                              C.POINTER(A.fw_stats)]
    lib.fw_render_scene.restype = C.c_int
    lib.fw_render_scene.argtypes = [C.POINTER(A.fw_scene_desc), C.POINTER(A.fw_render_params), C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.POINTER(A.fw_stats)]
    lib.fw_render_scene_tiled.restype = C.c_int
    lib.fw_render_scene_tiled.argtypes = [C.POINTER(A.fw_scene_desc), C.POINTER(A.fw_render_params), C.POINTER(C.c_int), C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(A.fw_stats)]
    lib.fw_render_progressive.restype = C.c_int
    lib.fw_render_progressive.argtypes = [C.c_void_p, C.POINTER(A.fw_render_params), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.POINTER(A.fw_stats)]
    lib.fw_release_workspace.restype = None
    lib.fw_release_workspace.argtypes = [C.c_int]
    lib.fw_selftest_arith.restype = C.c_int
    lib.fw_selftest_arith.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.fw_selftest_libm.restype = C.c_int
    lib.fw_selftest_libm.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fw_set_option.restype = C.c_int
    lib.fw_set_option.argtypes = [C.c_char_p, C.c_char_p]
    lib.fw_selftest_wide_bvh.restype = C.c_int
    lib.fw_selftest_wide_bvh.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.fw_selftest_bvh_build.restype = C.c_int
    lib.fw_selftest_bvh_build.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    lib.fw_init.restype = C.c_int
    lib.fw_init.argtypes = [C.c_int, C.c_uint64]
    if lib.fw_abi_version() != A.FW_ABI_VERSION:
        raise FireworkError(A.FW_ERR_BAD_ARG, "ABI version mismatch between _abi.py and libfirework_hip.so")
    _lib = lib
    if preload:
        init(device)
    return lib


def init(device=None, arena_bytes=0):
    """fw_init: explicit, idempotent initialisation of one device (default: LOCAL_RANK's, else 0).  arena_bytes 0 = the default
    arena, A.FW_INIT_NO_ARENA = none."""
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0") or 0)
        if device >= max(1, lib.fw_device_count()):
            device = 0
    _check(lib, lib.fw_init(int(device), int(arena_bytes)))


def _check(lib, st):
    if st != A.FW_OK:
        raise FireworkError(st, f"{lib.fw_strerror(st).decode()} | {lib.fw_last_error().decode()}")


def set_option(name, value=None):
    """fw_set_option: one runtime switch of the library (FIREWORK_<NAME>; the environment itself is read once, at load).
    value None = back to the default; name None = back to what the environment said at load time."""
    lib = load()
    _check(lib, lib.fw_set_option(None if name is None else str(name).encode(), None if value is None else str(value).encode()))


class options:
    """with _lib.options(FIREWORK_BVH="median", NO_DEFER="1"): ...  — switches set for the block, defaults restored after it."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_option(k, None)
        return False


def has_ab():
    """True for the A/B build (make ab: -DFW_AB=1), which also carries the measured-slower alternative kernels and their switches."""
    return hasattr(load(), "fw_debug_ab")


def selftest_wide_bvh(boxes, fmt):
    """fw_selftest_wide_bvh (CPU only): wide-node builder + invariant check over (n, 6) float32 item boxes.  -> (violations, stats)"""
    lib = load()
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    bad = C.c_uint32()
    stats = (C.c_uint32 * 4)()
    _check(lib, lib.fw_selftest_wide_bvh(b.ctypes.data, b.shape[0], int(fmt), C.byref(bad), stats))
    return int(bad.value), dict(nodes=int(stats[0]), leaves=int(stats[1]), free_slots=int(stats[2]), depth=int(stats[3]))


def selftest_bvh_build(boxes, threads):
    """fw_selftest_bvh_build (CPU only): the host tree builders with `threads` threads.  -> (hash of the median tree, hash of the SAH tree, stats)"""
    lib = load()
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    h = (C.c_uint64 * 2)()
    st = (C.c_uint32 * 4)()
    _check(lib, lib.fw_selftest_bvh_build(b.ctypes.data, b.shape[0], int(threads), h, st))
    return int(h[0]), int(h[1]), dict(median_nodes=int(st[0]), median_depth=int(st[1]), sah_nodes=int(st[2]), sah_depth=int(st[3]))


def selftest_arith(n, seed=1, mode=0, device=0):
    lib = load()
    d, s = C.c_uint64(), C.c_uint64()
    _check(lib, lib.fw_selftest_arith(device, n, seed, mode, C.byref(d), C.byref(s)))
    return int(d.value), int(s.value)


LIBM_FN = dict(logf=0, log10f=1, sinf=2, asinf=3, acosf=4, atanf=5, atan2f=6, powf=7)


def selftest_libm(fn, x, y=None, device=0):
    """fw_selftest_libm: the device's restated glibc function `fn` over float32 arrays."""
    lib = load()
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    yy = None if y is None else np.ascontiguousarray(y, np.float32)
    _check(lib, lib.fw_selftest_libm(device, LIBM_FN[fn], x.size, x.ctypes.data, None if yy is None else yy.ctypes.data, out.ctypes.data))
    return out


def build_id():
    """sha256 of the loaded library file: identifies the kernels a checkpoint's sums were made with."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def release_workspace(device=0):
    load().fw_release_workspace(device)


def device_count():
    return load().fw_device_count()


class DeviceScene:
    """An uploaded scene (`fw_scene*`): SoA scene arrays + TLAS/BLAS resident in HBM."""

    def __init__(self, scene_desc, device=0):
        lib = load()
        self._lib = lib
        self._desc = scene_desc  # keep host buffers alive
        h = C.c_void_p()
        _check(lib, lib.fw_scene_create(scene_desc.ptr(), device, C.byref(h)))
        self.handle = h
        self.device = device

    def render(self, renderer, pixel_ids=None, out_device_ptrs=None, stream=None):
        """fw_render.  out_device_ptrs = (rgb8, gamma, linear) raw device pointers (ints or None) to
        keep results in HBM (multi-GPU gather path); otherwise numpy host arrays are returned."""
        from .api import RenderResult
        lib = self._lib
        p = renderer.to_params(pixel_ids)
        n = int(pixel_ids.shape[0]) if pixel_ids is not None else p.width * p.height
        st = A.fw_stats()
        if out_device_ptrs is not None:
            p.outputs_on_device = 1
            p.stream = C.c_void_p(stream) if stream else None
            ptrs = [C.c_void_p(x) if x else None for x in out_device_ptrs]
            _check(lib, lib.fw_render(self.handle, C.byref(p), ptrs[0], ptrs[1], ptrs[2], C.byref(st)))
            return st.as_dict()
        rgb8 = np.empty((n, 3), np.uint8)
        gam = np.empty((n, 3), np.float32)
        lin = np.empty((n, 3), np.float32)
        _check(lib, lib.fw_render(self.handle, C.byref(p), rgb8.ctypes.data, gam.ctypes.data, lin.ctypes.data,
                                  C.byref(st)))
        return RenderResult(rgb8, gam, lin, st.as_dict(), p.width, p.height)

    def render_progressive(self, renderer, first_sample, accum, pixel_ids=None):
        """fw_render_progressive: adds samples [first_sample, first_sample + renderer's samples) to `accum`
        ((n_pixels, 4) float32, zeros before the first call; updated in place) and returns the image resolved so far."""
        from .api import RenderResult
        lib = self._lib
        p = renderer.to_params(pixel_ids)
        n = int(pixel_ids.shape[0]) if pixel_ids is not None else p.width * p.height
        assert accum.dtype == np.float32 and accum.shape == (n, 4) and accum.flags["C_CONTIGUOUS"]
        st = A.fw_stats()
        rgb8 = np.empty((n, 3), np.uint8)
        gam = np.empty((n, 3), np.float32)
        lin = np.empty((n, 3), np.float32)
        _check(lib, lib.fw_render_progressive(self.handle, C.byref(p), int(first_sample), accum.ctypes.data, rgb8.ctypes.data,
                                              gam.ctypes.data, lin.ctypes.data, C.byref(st)))
        return RenderResult(rgb8, gam, lin, st.as_dict(), p.width, p.height)

    def close(self):
        if self.handle:
            self._lib.fw_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_scene(scene_desc, renderer, pixel_ids=None, device=0):
    """fw_render_scene: the one-shot `Renderer::render(scene)` shape (conversion + BVH + render)."""
    from .api import RenderResult
    lib = load()
    p = renderer.to_params(pixel_ids)
    n = int(pixel_ids.shape[0]) if pixel_ids is not None else p.width * p.height
    rgb8 = np.empty((n, 3), np.uint8)
    gam = np.empty((n, 3), np.float32)
    lin = np.empty((n, 3), np.float32)
    st = A.fw_stats()
    _check(lib, lib.fw_render_scene(scene_desc.ptr(), C.byref(p), device, rgb8.ctypes.data, gam.ctypes.data,
                                    lin.ctypes.data, C.byref(st)))
    return RenderResult(rgb8, gam, lin, st.as_dict(), p.width, p.height)


def render_scene_tiled(scene_desc, renderer, devices):
    """fw_render_scene_tiled: one process, one host thread per listed device, tiles scattered into host buffers."""
    from .api import RenderResult
    lib = load()
    p = renderer.to_params(None)
    n = p.width * p.height
    rgb8 = np.empty((n, 3), np.uint8)
    gam = np.empty((n, 3), np.float32)
    lin = np.empty((n, 3), np.float32)
    st = A.fw_stats()
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    _check(lib, lib.fw_render_scene_tiled(scene_desc.ptr(), C.byref(p), devs, len(devices), rgb8.ctypes.data, gam.ctypes.data,
                                          lin.ctypes.data, C.byref(st)))
    return RenderResult(rgb8, gam, lin, st.as_dict(), p.width, p.height)
