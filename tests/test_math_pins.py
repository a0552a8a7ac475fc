"""The reference's tests of the scalar helpers under this path, restated -- against the oracle here, and (GPU-marked twins)
against the device's own copies through `mi_pt_math_probe`:

  FloatingPoint.NextUpDownFloat   src/tests/fp_tests.cpp:29-47     NextFloatUp / NextFloatDown (OffsetRayOrigin, EFloat)
  EFloat.{Add,Sub,Mul,Div}        src/tests/fp_tests.cpp:166-260   interval arithmetic under Sphere::Intersect's quadratic
  FindInterval.Basics             src/tests/find_interval.cpp:8    the search inside Distribution1D::SampleDiscrete

(EFloat.Abs / EFloat.Sqrt exercise operations Sphere::Intersect does not use: Quadratic takes the root of the discriminant in
double, efloat.h:271-290. tests/bounds.cpp covers Bounds iterators, Distance and Union, none of which is on the render path:
the BVH build's use of Union is pinned by the reference's node counts, tests/test_frontend.py.)
The reference draws its operands from its PCG32 stream; containment does not depend on which operands, so these use numpy's
seeded generator with the reference's distributions (exponent uniform in [-6, 6]; no error / <= 1024 ulp / <= 2^20 ulp / up to
4 |v|) and its adversarial choice of the precise value (an end of the interval two times in three)."""
import ctypes as C

import numpy as np
import pytest

_F = C.POINTER(C.c_float)


def _oracle_probe(ob):
    lib = ob.lib()
    lib.oracle_math_probe.argtypes = [C.c_int, C.c_uint32, _F, _F, _F]

    def probe(op, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(x if y is None else y, np.float32)
        out = np.zeros((len(x), 3), np.float32)
        lib.oracle_math_probe(op, len(x), x.ctypes.data_as(_F), y.ctypes.data_as(_F), out.ctypes.data_as(_F))
        return out
    return probe


def _device_probe(pt):
    return lambda op, x, y=None: pt.math_probe(op, x, y)


def _next_up_down(probe):
    inf = np.float32(np.inf)
    special = np.array([[-0.0, 0], [0.0, 0], [inf, 0], [-inf, 0]], np.float32)
    r = probe(0, special)
    assert r[0, 0] > 0 and r[1, 1] < 0                         # NextFloatUp(-0.f) > 0, NextFloatDown(0.f) < 0
    assert r[2, 0] == inf and r[2, 1] < inf                    # up(inf) == inf, down(inf) < inf
    assert r[3, 1] == -inf and r[3, 0] > -inf
    rng = np.random.default_rng(1)
    f = rng.integers(0, 1 << 32, 100000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    f = f[np.isfinite(f)]
    r = probe(0, np.stack([f, np.zeros_like(f)], 1))
    assert np.array_equal(r[:, 0], np.nextafter(f, inf)) and np.array_equal(r[:, 1], np.nextafter(f, -inf))


def _efloats(rng, n):
    """getFloat, tests/fp_tests.cpp:108-136: (value, error bound) pairs and the interval they stand for."""
    val = (10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    kind = rng.integers(0, 4, n)
    bits = val.view(np.uint32).astype(np.uint64)
    small = (bits + rng.integers(0, 1024, n).astype(np.uint64)).astype(np.uint32).view(np.float32)
    big = (bits + rng.integers(0, 1 << 20, n).astype(np.uint64)).astype(np.uint32).view(np.float32)
    err = np.select([kind == 0, kind == 1, kind == 2], [np.zeros(n, np.float32), np.abs(small - val), np.abs(big - val)],
                    (4 * rng.random(n).astype(np.float32)) * np.abs(val)).astype(np.float32)
    v = (np.where(rng.random(n) < .5, -1, 1) * val).astype(np.float32)
    inf = np.float32(np.inf)
    low = np.where(err == 0, v, np.nextafter((v - err).astype(np.float32), -inf))    # EFloat(v, err), efloat.h:52-62
    high = np.where(err == 0, v, np.nextafter((v + err).astype(np.float32), inf))
    return np.stack([v, err], 1), low.astype(np.float64), high.astype(np.float64)


def _precise(rng, low, high):
    """getPrecise, tests/fp_tests.cpp:140-158."""
    t = rng.random(len(low)).astype(np.float32).astype(np.float64)
    mid = np.clip((1 - t) * low + t * high, low, high)
    return np.select([(k := rng.integers(0, 3, len(low))) == 0, k == 1], [low, high], mid)


def _efloat_ops(probe, n=200000):
    rng = np.random.default_rng(7)
    a, alo, ahi = _efloats(rng, n)
    b, blo, bhi = _efloats(rng, n)
    pa, pb = _precise(rng, alo, ahi), _precise(rng, blo, bhi)
    with np.errstate(all="ignore"):
        for op, fn in ((1, np.add), (2, np.subtract), (3, np.multiply), (4, np.divide)):
            r = probe(op, a, b).astype(np.float64)
            want = fn(pa, pb).astype(np.float32).astype(np.float64)   # `float preciseResult = precise[0] op precise[1]`
            ok = np.ones(n, bool)
            if op == 4:   # the denominator's interval must not straddle zero nor be wide against its centre (fp_tests.cpp:249-252)
                ok = ~((blo * bhi < 0) | ((bhi - blo) / 2 > .25 * np.abs(blo)))
                assert ok.mean() > .3
            assert (want[ok] >= r[ok, 1]).all() and (want[ok] <= r[ok, 2]).all(), op
            assert np.array_equal(r[:, 0], fn(a[:, 0], b[:, 0]).astype(np.float32).astype(np.float64), equal_nan=True)   # the value itself is the plain float result


def _find_interval(probe):
    q = lambda v: int(probe(5, np.array([[v, 0]], np.float32))[0, 0])
    assert q(-1) == 0 and q(100) == 8            # clamped to [0, size - 2]
    for i in range(9):
        assert q(i) == i and q(i + .5) == i
        if i > 0:
            assert q(i - .5) == i - 1


def test_next_float_up_down(ob):
    _next_up_down(_oracle_probe(ob))


def test_efloat_operations_contain_the_precise_result(ob):
    _efloat_ops(_oracle_probe(ob))


def test_find_interval_basics(ob):
    _find_interval(_oracle_probe(ob))


@pytest.mark.gpu
def test_scalar_helpers_on_the_device(pt, ob):
    """The same three reference tests over the device's NextFloatUp / NextFloatDown, EFloat and SampleDiscrete
    (mi_pt_math_probe), and the device's results equal to the oracle's bit for bit on the same operands."""
    dev, orc = _device_probe(pt), _oracle_probe(ob)
    _next_up_down(dev)
    _efloat_ops(dev)
    _find_interval(dev)
    rng = np.random.default_rng(11)
    a, _, _ = _efloats(rng, 50000)
    b, _, _ = _efloats(rng, 50000)
    for op in (1, 2, 3, 4):
        assert np.array_equal(dev(op, a, b).view(np.int32), orc(op, a, b).view(np.int32)), op
