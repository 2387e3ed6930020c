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
