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
