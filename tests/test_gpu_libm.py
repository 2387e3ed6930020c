"""The device's restated glibc functions (fw_selftest_libm) against the host's libm, bit for bit (-m gpu).
Inputs: what the renderer produces (unit-interval draws, directions, texture coordinates, radiance) plus random bit patterns."""
import numpy as np
import pytest

from firework_amd import _lib

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same(a, b):
    return (_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))


def _rand_bits(rng, n):
    return rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("fn,gen", [
    ("log10f", lambda r, n: np.concatenate([np.arange(1 << 24, dtype=np.float32) * np.float32(2.0 ** -24), _rand_bits(r, n)])),   # every xi the RNG can draw
    ("logf", lambda r, n: np.abs(_rand_bits(r, n))),
    ("sinf", lambda r, n: np.concatenate([(r.random(n, np.float32) - 0.5) * np.float32(2e4), (r.random(n, np.float32) - 0.5) * 8, _rand_bits(r, n)])),
    ("asinf", lambda r, n: np.concatenate([r.random(n, np.float32) * 2 - 1, _rand_bits(r, n)])),
    ("acosf", lambda r, n: np.concatenate([r.random(n, np.float32) * 2 - 1, _rand_bits(r, n)])),
    ("atanf", lambda r, n: np.concatenate([(r.random(n, np.float32) - 0.5) * 64, _rand_bits(r, n)])),
])
def test_one_argument_functions(oracle, fn, gen):
    rng = np.random.default_rng(1234)
    x = np.ascontiguousarray(gen(rng, 1 << 21), np.float32)
    dev, host = _lib.selftest_libm(fn, x), oracle.libm(fn, x)
    bad = ~_same(dev, host)
    assert not bad.any(), (fn, int(bad.sum()), x[bad][:4], dev[bad][:4], host[bad][:4])


def test_atan2f(oracle):
    rng = np.random.default_rng(5)
    n = 1 << 22
    y = np.concatenate([rng.random(n, np.float32) * 2 - 1, _rand_bits(rng, n)])
    x = np.concatenate([rng.random(n, np.float32) * 2 - 1, _rand_bits(rng, n)])
    dev, host = _lib.selftest_libm("atan2f", y, x), oracle.libm("atan2f", y, x)
    assert _same(dev, host).all()


def test_powf(oracle):
    rng = np.random.default_rng(6)
    n = 1 << 21
    base = np.concatenate([rng.random(n, np.float32), rng.random(n, np.float32) * 64, _rand_bits(rng, n)])
    for e in (5.0, 1.0 / 2.2, 0.5, 2.4):                       # schlick (util.rs:69-73); gamma 2.2 / 2.0 (render.rs:186)
        y = np.full_like(base, np.float32(e))
        assert _same(_lib.selftest_libm("powf", base, y), oracle.libm("powf", base, y)).all(), e
    y = _rand_bits(rng, base.size)
    assert _same(_lib.selftest_libm("powf", base, y), oracle.libm("powf", base, y)).all()
