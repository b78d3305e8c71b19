"""GPU: include/mirres_fmath.h returns the SAME BITS on gfx950 as on the host, argument by argument — the premise of the per-pixel parity tests.
Device side: mirres_fmath_checksum (the header compiled by hipcc into libmirres.so, with the short square root of the shading kernels plugged in);
host side: oracle/libfmathcheck.so (the same header compiled by g++).  The checksum is an order-free sum over (argument bits, result bits), so one
differing result anywhere in a range changes it.  With >= 32 host cores every one of the 2^32 arguments of each one-argument function is covered
(~2 s per function); with fewer, 2^29 arguments in blocks spread over the whole bit range."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "sin", 1: "cos", 2: "acos", 3: "exp", 4: "exp2", 5: "pow5", 6: "pow8", 7: "pow128", 8: "sigmoid", 16: "atan2", 17: "short division", 18: "short sqrt"}


@pytest.fixture(scope="module")
def both(oracle):
    import torch
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    H = C.CDLL(os.path.join(ROOT, "oracle", "libfmathcheck.so"))
    H.fmath_checksum.restype = C.c_uint64
    H.fmath_checksum.argtypes = [C.c_int, C.c_uint32, C.c_uint64]
    H.fmath_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]

    def dev(fn, first, count):
        out = C.c_uint64()
        check(lib().mirres_fmath_checksum(fn, first, count, C.byref(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fmath_checksum")
        return out.value
    return H, dev, torch


def _ranges():
    if (os.cpu_count() or 1) >= 32 or os.environ.get("MIRRES_FMATH_FULL", "0") == "1":
        return [(0, 1 << 32)]
    return [(b << 26, 1 << 23) for b in range(64)]      # 64 blocks of 2^23 arguments, one per 2^26 bit patterns: every exponent, both signs


@pytest.mark.parametrize("fn", [0, 1, 2, 3, 4, 5, 6, 7, 8, 16])
def test_device_and_host_return_the_same_bits(both, fn):
    H, dev, _ = both
    for first, count in _ranges():
        d, h = dev(fn, first, count), H.fmath_checksum(fn, first, count)
        if d != h:          # narrow it down for the message
            lo, n = first, count
            while n > 1:
                half = n // 2
                if dev(fn, lo, half) != H.fmath_checksum(fn, lo, half):
                    n = half
                else:
                    lo, n = lo + half, n - half
            pytest.fail("%s: device and host differ at argument bits 0x%08x (%r)" % (NAMES[fn], lo, float(np.uint32(lo).view(np.float32))))


def test_short_division_and_square_root_where_the_header_uses_them(both):
    """MRF_SQRT is the shading kernels' short sequence on the device (acos: argument in [2^-25, 1/2] or zero); the short division is compared with
    IEEE division for divisors 2^-60 .. 2^60 and numerators 2^-20 .. 2^20 (the range the sequence is proved for is 2^+-102, mirres_selfcheck_arith)."""
    H, dev, _ = both
    b = lambda f: int(np.float32(f).view(np.uint32))
    assert dev(18, b(2.0 ** -26), b(1.0) - b(2.0 ** -26)) == H.fmath_checksum(18, b(2.0 ** -26), b(1.0) - b(2.0 ** -26))
    assert dev(18, 0, 1) == H.fmath_checksum(18, 0, 1)
    for sign in (0, 0x80000000):
        lo, hi = b(2.0 ** -60), b(2.0 ** 60)
        assert dev(17, sign + lo, hi - lo) == H.fmath_checksum(17, sign + lo, hi - lo)


def test_eval_entry_point_matches_host_on_path_like_arguments(both):
    """mirres_fmath_eval on the arguments the path feeds: unit-vector components, angles, EAW exponents — element-wise bit equality."""
    H, dev, torch = both
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    rng = np.random.default_rng(5)
    n = 1 << 20
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    cases = [(0, rng.uniform(-7, 7, n), None), (1, rng.uniform(-7, 7, n), None), (2, v[:, 1], None), (3, -rng.exponential(5.0, n), None), (8, rng.normal(0, 4, n), None),
             (5, rng.uniform(0, 1, n), None), (6, rng.uniform(0, 1, n), None), (16, v[:, 2], v[:, 0])]
    for fn, a, b_ in cases:
        a = np.ascontiguousarray(a, np.float32); b_ = None if b_ is None else np.ascontiguousarray(b_, np.float32)
        want = np.empty_like(a)
        H.fmath_eval(fn, a.ctypes.data, None if b_ is None else b_.ctypes.data, want.ctypes.data, n)
        ta = torch.from_numpy(a).cuda(); tb = None if b_ is None else torch.from_numpy(b_).cuda(); to = torch.empty_like(ta)
        check(lib().mirres_fmath_eval(fn, C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()) if tb is not None else None, C.c_void_p(to.data_ptr()), n,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fmath_eval")
        assert np.array_equal(to.cpu().numpy().view(np.uint32), want.view(np.uint32)), NAMES[fn]
    assert lib().mirres_fmath_eval(9, None, None, None, 4, None) != 0 and lib().mirres_fmath_checksum(3, 0, 1 << 33, None, None) != 0
