"""The compiled-language host side: include/firework.hpp (C++ mirror of the reference's builder API) +
examples/cornell_box.cpp (examples/cornell_box.rs line for line), linked against the C ABI only."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "cornell_box")


def _build():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", ROOT, "examples"])


def test_cpp_example_builds_and_has_no_cpu_fallback():
    _build()
    from firework_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    p = subprocess.run([EXE, "8", "8", "1"], capture_output=True, text=True)
    assert p.returncode == 1 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_cpp_host_renders_the_same_bytes_as_the_python_host():
    _build()
    out = subprocess.check_output([EXE, "64", "64", "16"], text=True)
    assert "Finished Rendering in" in out
    m = re.search(r"rays=(\d+) samples=(\d+) .*fnv1a=([0-9a-f]{16})", out)
    from firework_amd import scenes
    s, r = scenes.cornell_box()
    res = r.width(64).height(64).samples(16).render_full(s)
    h = 1469598103934665603
    for b in res.rgb8.reshape(-1).tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(m.group(1)) == res.stats["rays"] and int(m.group(2)) == res.stats["samples"]
    assert m.group(3) == f"{h:016x}"


@pytest.mark.gpu
def test_cpp_progressive_render_ends_bit_identical():
    """include/firework.hpp's render_progressive (fw_render_progressive through the C ABI): 3 passes, same image hash."""
    _build()
    one = subprocess.run([EXE, "40", "32", "12"], capture_output=True, text=True, timeout=120)
    three = subprocess.run([EXE, "40", "32", "12", "/dev/null", "3"], capture_output=True, text=True, timeout=120)
    assert one.returncode == 0 and three.returncode == 0, one.stderr + three.stderr
    h1 = re.search(r"fnv1a=([0-9a-f]{16})", one.stdout).group(1)
    h3 = re.search(r"fnv1a=([0-9a-f]{16})", three.stdout).group(1)
    assert h1 == h3 and three.stdout.count("pass ") == 3
