"""CPU checks of the host-side tree builders (fw_selftest_bvh_build): the parallel builds of scene creation must reproduce the sequential
recursion bit for bit — the median-split tree is the reference's (bvh.rs:21-71: stable sort of the sub-slice at every level, split at n/2,
leaves of one or two items) and fixes tie ranks and gate boxes; the SAH tree is what the device walks."""
import time

import numpy as np
import pytest

from firework_amd import _lib


def _boxes(n, seed, flat=False):
    r = np.random.default_rng(seed)
    c = r.uniform(-10, 10, (n, 3)).astype(np.float32)
    if flat:
        c[:, 1] = np.float32(0.25)                      # equal centres along an axis: the stable sort's order decides the tree
        c[: n // 3, 0] = np.float32(1.5)
    e = r.uniform(0.001, 0.3, (n, 3)).astype(np.float32)
    return np.concatenate([c - e, c + e], axis=1)


def test_known_topologies():
    """N = 8 -> 7 nodes; N = 968 -> 1 023 nodes, depth 9; N = 1 409 -> 1 793 nodes, depth 10."""
    for n, nodes, depth in ((8, 7, 2), (968, 1023, 9), (1409, 1793, 10)):
        _, _, st = _lib.selftest_bvh_build(_boxes(n, n), 1)
        assert (st["median_nodes"], st["median_depth"]) == (nodes, depth)


@pytest.mark.parametrize("n,flat", [(1, False), (2, False), (3, False), (5000, False), (70001, False), (70001, True), (300000, False)])
def test_parallel_builds_equal_the_sequential_ones(n, flat):
    b = _boxes(n, 7 * n + flat, flat)
    h1 = _lib.selftest_bvh_build(b, 1)
    for threads in (2, 5, 8):
        assert _lib.selftest_bvh_build(b, threads) == h1, (n, threads)


def test_parallel_build_is_faster_where_there_are_cores():
    import os
    if (os.cpu_count() or 1) < 4:
        pytest.skip("needs a few cores")
    b = _boxes(400000, 3)
    t = []
    for threads in (1, 8):
        t0 = time.perf_counter(); _lib.selftest_bvh_build(b, threads); t.append(time.perf_counter() - t0)
    print(f"400 000 boxes, median + SAH tree: {t[0] * 1e3:.0f} ms on 1 thread, {t[1] * 1e3:.0f} ms on 8")
    assert t[1] < t[0]


def test_nan_centre_is_an_error_code():
    b = _boxes(100, 1)
    b[17, 0] = np.nan; b[17, 3] = np.nan
    with pytest.raises(_lib.FireworkError) as e:
        _lib.selftest_bvh_build(b, 4)
    from firework_amd import _abi as A
    assert e.value.status == A.FW_ERR_NAN_BBOX
