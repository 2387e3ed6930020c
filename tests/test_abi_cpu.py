"""CPU-side checks of the product library: it loads, exports every symbol include/firework_hip.h declares,
and refuses to compute without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from firework_amd import _abi as A
from firework_amd import _lib, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "firework_hip.h")).read()
    text = text[text.index("/* ---- entry points"):]
    return sorted(set(re.findall(r"\b(fw_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = _lib.load()
    names = _declared_symbols()
    assert {"fw_render", "fw_render_scene", "fw_scene_create", "fw_scene_destroy", "fw_device_count", "fw_strerror",
            "fw_last_error", "fw_abi_version"} <= set(names)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.fw_abi_version() == A.FW_ABI_VERSION


def test_struct_sizes_match_header():
    """ctypes mirror vs the C compiler's layout of include/firework_hip.h."""
    import subprocess
    import tempfile
    src = '#include "firework_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(fw_texture),sizeof(fw_material),sizeof(fw_shape),sizeof(fw_object),sizeof(fw_environment),' \
          'sizeof(fw_scene_desc),sizeof(fw_camera_settings),sizeof(fw_render_params),sizeof(fw_stats));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).split()
    got = [C.sizeof(x) for x in (A.fw_texture, A.fw_material, A.fw_shape, A.fw_object, A.fw_environment,
                                 A.fw_scene_desc, A.fw_camera_settings, A.fw_render_params, A.fw_stats)]
    assert got == [int(x) for x in out]


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    s, r = scenes.cornell_box()
    with pytest.raises(_lib.FireworkError) as e:
        r.width(8).height(8).samples(1).render(s)
    assert e.value.status == A.FW_ERR_NO_DEVICE


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under firework_amd/ may import, link or name it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "firework_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle_binding" not in text and "fwo_" not in text, f


_SHIM = r"""
#include <stdio.h>
#include <stdlib.h>
static void note(const char *n) { const char *f = getenv("FW_SHIM_LOG"); if (f) { FILE *p = fopen(f, "a"); if (p) { fprintf(p, "%s\n", n); fclose(p); } } }
#define TRAP(name) int name() { note(#name); return 100; }
TRAP(hipInit) TRAP(hipGetDeviceCount) TRAP(hipSetDevice) TRAP(hipGetDevice) TRAP(hipMalloc) TRAP(hipHostMalloc) TRAP(hipMemGetInfo)
TRAP(hipStreamCreateWithFlags) TRAP(hipEventCreateWithFlags) TRAP(hipGetDevicePropertiesR0600) TRAP(hipGetDeviceProperties)
TRAP(hipLaunchKernel) TRAP(hipFuncGetAttributes) TRAP(hipMemcpy) TRAP(hipMemcpyAsync) TRAP(hipDeviceSynchronize)
"""


def test_loading_the_library_makes_no_hip_call(tmp_path):
    """ABI v7 / SURVEY §8(b) "Ownership": dlopen alone creates no context, queries no device and allocates nothing — the first HIP call
    is made by fw_init or by the first call that needs the device.  A preloaded shim traps the runtime's entry points."""
    import subprocess
    import sys
    (tmp_path / "shim.c").write_text(_SHIM)
    shim = str(tmp_path / "shim.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-w", "-o", shim, str(tmp_path / "shim.c")])
    log = tmp_path / "calls.txt"
    code = ("import ctypes, os; lib = ctypes.CDLL(os.environ['FW_LIB_PATH']); print('loaded', lib.fw_abi_version());"
            "open(os.environ['FW_SHIM_LOG'], 'a').write('-- after load\\n'); lib.fw_init.argtypes = [ctypes.c_int, ctypes.c_uint64]; print('init', lib.fw_init(0, 0))")
    env = dict(os.environ, LD_PRELOAD=shim, FW_SHIM_LOG=str(log), FW_LIB_PATH=_lib.LIB_PATH, LOCAL_RANK="0")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True)
    assert "loaded %d" % A.FW_ABI_VERSION in out
    before, after = log.read_text().split("-- after load\n")
    assert before == "", "HIP calls during dlopen: " + before
    assert "hipGetDeviceCount" in after          # ... and fw_init is where the runtime is first asked for a device


def test_fw_init_without_a_device_fails_loudly():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    lib = _lib.load()
    assert lib.fw_init(0, 0) == A.FW_ERR_NO_DEVICE
    with pytest.raises(_lib.FireworkError):
        _lib.init()


def test_every_option_the_header_names_is_known_to_the_library_and_nothing_else():
    """fw_set_option (ABI v6+) without a device: each switch named in the header's list is accepted (value NULL = back to its default), a
    name the build does not know is FW_ERR_BAD_ARG with the name in fw_last_error() — the list in include/firework_hip.h stays the list."""
    import re
    from firework_amd import _abi as A, _lib
    text = open(os.path.join(ROOT, "include", "firework_hip.h")).read()
    block = text[text.index("except the diagnostics NO_EXACT / EXACT_ALL.  Names:"):text.index("Returns FW_ERR_BAD_ARG for a name this build does not know")]
    names = set()
    for line in block.split("Names:")[1].splitlines():          # "     NAME=value, OTHER_NAME      what it does": the names are the first column
        first = re.split(r"\s{2,}", line.strip())[0]
        if re.match(r"[A-Z][A-Z0-9_]{2,}", first):
            names |= set(re.findall(r"\b[A-Z][A-Z0-9_]{2,}\b", first))
    names = sorted(names)
    assert {"BVH", "WIDE", "STREAMS", "EXACT_PRODUCT", "PHASE_LOCK", "GRAPH", "TRACE", "DUMP_PATH", "NO_CHAIN"} <= set(names), names
    lib = _lib.load()
    for n in names:
        assert lib.fw_set_option(n.encode(), None) == A.FW_OK, n
    assert lib.fw_set_option(b"NO_SUCH_SWITCH", b"1") == A.FW_ERR_BAD_ARG
    assert b"NO_SUCH_SWITCH" in lib.fw_last_error()
    assert lib.fw_set_option(None, None) == A.FW_OK          # back to what the environment said at load time
