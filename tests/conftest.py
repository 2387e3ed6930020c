import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """The driver runs __graft_entry__.build() first; build here too if someone runs pytest on a fresh checkout."""
    import subprocess
    need = [os.path.join(ROOT, "firework_amd", "lib", "libfirework_hip.so"), os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "examples", "cornell_box")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "all"])


def pytest_sessionstart(session):
    _ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU oracle renders")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so); built on demand when g++ is available."""
    from oracle import oracle_binding as ob
    if not os.path.exists(ob.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    ob.load()
    return ob


def block_means(img, rows, cols):
    import numpy as np
    h, w, _ = img.shape
    bh, bw = h // rows, w // cols
    return np.array([[img[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw].reshape(-1, 3).mean(0) for c in range(cols)]
                     for r in range(rows)])


@pytest.fixture
def monkeypatch(monkeypatch):
    """The library reads FIREWORK_* from the environment once, when it is loaded; afterwards a switch is changed through
    fw_set_option.  Tests keep writing monkeypatch.setenv / delenv: this wrapper passes every FIREWORK_* name on to the loaded
    library as well (child processes still see the environment) and restores the defaults after the test."""
    from firework_amd import _lib
    process_only = {"FIREWORK_LIB", "FIREWORK_NO_TORCH"}
    touched = {}
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def push(name, value, before):
        if name.startswith("FIREWORK_") and name not in process_only and os.path.exists(_lib.LIB_PATH):
            try:
                _lib.set_option(name, value)
                touched.setdefault(name, before)
            except _lib.FireworkError:
                if value is not None:
                    pytest.skip(f"{name} is a switch of the A/B build (make ab; FIREWORK_LIB=firework_amd/lib/variants/lib_ab.so)")

    def setenv(name, value, prepend=None):
        before = os.environ.get(name)
        orig_set(name, value, prepend)
        push(name, str(value), before)

    def delenv(name, raising=True):
        before = os.environ.get(name)
        orig_del(name, raising)
        push(name, None, before)

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch
    for name, before in touched.items():
        _lib.set_option(name, before)
