"""CPU: include/mirres_fmath.h — the transcendental functions the HIP product and the oracle share — stays within 2 ulp of the correctly rounded
functions (double-precision glibc as the exact value).  The default run sweeps every binary32 argument of the ranges the path actually feeds
(angles within [-2 pi, 2 pi], cosines in [-1, 1], EAW / sigmoid exponents) plus blocks of the outer ranges; MIRRES_FMATH_FULL=1 sweeps each whole domain
(profiles/r03_fmath_accuracy.txt holds that run: sin 1.56, cos 1.56, acos 1.12, exp 1.01, exp2 0.95, pow5 1.37, x^8 0.50, x^128 0.99, atan2 1.72)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.environ.get("MIRRES_FMATH_FULL", "0") == "1"


@pytest.fixture(scope="module")
def chk(oracle):
    L = C.CDLL(os.path.join(ROOT, "oracle", "libfmathcheck.so"))
    L.fmath_max_ulp.restype = C.c_double
    L.fmath_max_ulp.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32)]
    L.fmath_atan2_max_ulp.restype = C.c_double
    L.fmath_atan2_max_ulp.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.fmath_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    return L


def bits(f):
    return int(np.float32(f).view(np.uint32))


def sweep(L, fn, lo, hi, both_signs=True):
    """max ulp error over the bit patterns of [lo, hi] (lo, hi >= 0) and, if asked, of [-hi, -lo]"""
    worst = 0.0
    for sign in ((0, 0x80000000) if both_signs else (0,)):
        wb = C.c_uint32()
        worst = max(worst, L.fmath_max_ulp(fn, sign + bits(lo), bits(hi) - bits(lo) + 1, C.byref(wb)))
    return worst


FN = dict(sin=0, cos=1, acos=2, exp=3, exp2=4, pow5=5, pow8=6, pow128=7, sigmoid=8)


@pytest.mark.parametrize("name,lo,hi,both,bound", [
    ("sin", 2.0 ** -12, 8.0, True, 2.0), ("cos", 2.0 ** -12, 8.0, True, 2.0),
    ("sin", 4096.0, 4224.0, True, 2.0), ("cos", 4096.0, 4224.0, True, 2.0),
    ("acos", 0.25, 1.0, True, 2.0), ("acos", 0.0, 2.0 ** -100, True, 2.0),
    ("exp", 0.03125, 16.0, True, 2.0), ("exp", 80.0, 88.8, False, 2.0), ("exp2", 0.03125, 16.0, True, 2.0),
    ("pow5", 2.0 ** -6, 1.0, False, 2.0), ("pow8", 2.0 ** -4, 1.0, False, 2.0), ("pow128", 0.75, 1.0, False, 2.0),
    ("sigmoid", 0.03125, 32.0, True, 3.0),   # 1 / (1 + exp(-x)): three roundings on top of exp's error (the formula torch.sigmoid restates)
])
def test_one_argument_functions_within_two_ulp(chk, name, lo, hi, both, bound):
    assert sweep(chk, FN[name], lo, hi, both) <= bound


@pytest.mark.skipif(not FULL, reason="whole-domain sweeps (minutes on 8 cores): MIRRES_FMATH_FULL=1")
@pytest.mark.parametrize("name,lo,hi,both,bound", [
    ("sin", 0.0, 8192.0, True, 2.0), ("cos", 0.0, 8192.0, True, 2.0), ("acos", 0.0, 1.0, True, 2.0), ("exp", 0.0, 88.8, False, 2.0),
    ("exp", 0.0, 87.3, True, 2.0), ("exp2", 0.0, 126.0, True, 2.0), ("pow5", 1e-4, 1.0, False, 2.0), ("pow8", 1e-4, 1.0, False, 2.0), ("pow128", 0.51, 1.0, False, 2.0)])
def test_whole_domains(chk, name, lo, hi, both, bound):
    assert sweep(chk, FN[name], lo, hi, both) <= bound


def test_atan2_within_two_ulp(chk):
    wy, wx = C.c_uint32(), C.c_uint32()
    n = 1 << (32 if FULL else 27)
    # the arctangent core with the octant fold (y / 1, every y in [2^-8, 2^8]), pseudo-random pairs of both signs, unit vectors around the circle
    assert chk.fmath_atan2_max_ulp(0, bits(2.0 ** -8), bits(2.0 ** 8) - bits(2.0 ** -8), C.byref(wy), C.byref(wx)) <= 2.0
    assert chk.fmath_atan2_max_ulp(1, 12345, n, C.byref(wy), C.byref(wx)) <= 2.0
    assert chk.fmath_atan2_max_ulp(2, 0, n, C.byref(wy), C.byref(wx)) <= 2.0 if FULL else True
    assert chk.fmath_atan2_max_ulp(2, 0x40000000, n, C.byref(wy), C.byref(wx)) <= 2.0


def test_special_values(chk):
    def ev(fn, a, b=None):
        a = np.asarray(a, np.float32); out = np.empty_like(a)
        bb = np.asarray(b, np.float32) if b is not None else None
        chk.fmath_eval(fn, a.ctypes.data, bb.ctypes.data if bb is not None else None, out.ctypes.data, a.size)
        return out
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    assert np.array_equal(ev(0, [0.0, -0.0]).view(np.uint32), np.array([0, 0x80000000], np.uint32))            # sin(+-0) = +-0
    assert np.isnan(ev(0, [inf, nan, 1e6])).all() and np.isnan(ev(1, [inf, nan, -1e6])).all()                # outside the stated domain: NaN, loudly
    assert ev(1, [0.0])[0] == 1.0
    assert np.array_equal(ev(2, [1.0, -1.0, 0.0]), np.array([0.0, np.float32(np.pi), np.float32(np.pi / 2)], np.float32))
    assert np.isnan(ev(2, [1.0000001, -1.5, nan])).all()
    assert np.array_equal(ev(3, [0.0, -200.0, 100.0, -inf, inf]), np.array([1.0, 0.0, inf, 0.0, inf], np.float32)) and np.isnan(ev(3, [nan])).all()
    assert np.array_equal(ev(4, [0.0, 1.0, -1.0, 10.0, -126.0, -149.0, -151.0, 128.0]), np.array([1, 2, 0.5, 1024, 2.0 ** -126, 2.0 ** -149, 0, inf], np.float32))
    pi = np.float32(np.pi)
    got = ev(16, [0.0, -0.0, 0.0, -0.0, 1.0, -1.0, inf, inf, 1.0], [1.0, 1.0, -1.0, -1.0, 0.0, 0.0, inf, -inf, nan])
    want = np.array([0.0, -0.0, pi, -pi, pi / 2, -pi / 2, pi / 4, 3 * np.float32(np.pi / 4), nan], np.float32)
    assert np.array_equal(got[:8].view(np.uint32), want[:8].view(np.uint32)) and np.isnan(got[8])
    assert np.array_equal(ev(5, [0.0, 1.0, 0.5]), np.array([0, 1, 2.0 ** -5], np.float32))
    assert np.array_equal(ev(6, [0.0, 1.0, 0.5]), np.array([0, 1, 2.0 ** -8], np.float32)) and np.array_equal(ev(7, [1.0, 0.5]), np.array([1, 2.0 ** -128], np.float32))
    assert np.array_equal(ev(8, [0.0, 200.0, -200.0]), np.array([0.5, 1.0, 0.0], np.float32))
