"""GPU: the short division / reciprocal / square-root sequences of the shading kernels (csrc/device_math.hpp, MR_LEAN_FP) return the bits of the
IEEE-754 operations — checked by the library's own self-check entry point on the code that ships (the frame tests compare against an oracle
that divides with the host's IEEE division, so a wrong last bit here would surface there only as a rare changed reservoir decision)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def test_division_and_square_root_sequences_are_exact():
    import torch
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    out = (C.c_uint64 * 4)()
    # 2^14 significands of b spread over [1, 2) + the last 256, each against ALL 2^23 significands of a (1.4e11 quotients, < 1 s);
    # every reciprocal of [1, 2) and every square root of [1, 4). The full 2^46 run (log2_b = 23, ~50 s): profiles/r02_div_exhaustive.txt.
    check(lib().mirres_selfcheck_arith(14, out, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "selfcheck")
    pairs, bad_div, bad_rcp, bad_sqrt = list(out)
    assert pairs == (2 ** 14 + 256) * 2 ** 23
    assert (bad_div, bad_rcp, bad_sqrt) == (0, 0, 0)


def test_selfcheck_rejects_bad_arguments():
    from mirres_restir_nerf_mesh_amd._lib import lib
    out = (C.c_uint64 * 4)()
    assert lib().mirres_selfcheck_arith(3, out, None) != 0 and lib().mirres_selfcheck_arith(24, out, None) != 0
    assert lib().mirres_selfcheck_arith(12, None, None) != 0
