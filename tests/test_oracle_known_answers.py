"""CPU: the oracle against known-answer values derived from the reference's formulas and against independent
numpy restatements. The reference has no tests/golden vectors of its own for this path (parity unpinned)."""
import numpy as np


def test_rng_known_answers(oracle):
    s = oracle.seed(0, 0, 0)
    assert s == 0x741c187d
    vals = []
    for _ in range(3):
        v, s = oracle.next1d(s); vals.append(v)
    np.testing.assert_allclose(vals, [0.71806329, 0.61330026, 0.92872769], rtol=0, atol=5e-9)
    assert oracle.seed(1, 2, 3) == 0x61c09a34 and oracle.seed(799, 799, 12345) == 0xfdd3ab5f
    # coordinates are masked to 16 bits (random.slang:4-5)
    assert oracle.seed(65536 + 5, 9, 1) == oracle.seed(5, 9, 1)


def _tea_py(px, py, n):
    def il(v):
        v &= 0xffff
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    M = 0xffffffff
    v0, v1, s = (il(px) | (il(py) << 1)) & M, n & M, 0
    for _ in range(16):
        s = (s + 0x9e3779b9) & M
        v0 = (v0 + ((((v1 << 4) + 0xa341316c) & M) ^ ((v1 + s) & M) ^ (((v1 >> 5) + 0xc8013ea4) & M))) & M
        v1 = (v1 + ((((v0 << 4) + 0xad90777d) & M) ^ ((v0 + s) & M) ^ (((v0 >> 5) + 0x7e95761e) & M))) & M
    return v0


def test_rng_vs_python_restatement(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        x, y, n = (int(v) for v in rng.integers(0, 2**32, 3))
        assert oracle.seed(x, y, n) == _tea_py(x, y, n)
    st = 12345
    for _ in range(50):
        v, st2 = oracle.next1d(st)
        assert st2 == (1664525 * st + 1013904223) & 0xffffffff and v == np.float32((st2 >> 8) * 2.0**-24)
        st = st2


def test_morton(oracle):
    assert oracle.expand_bits(1023) == 0x09249249
    assert oracle.morton3d(0.5, 0.5, 0.5) == 939524096 and oracle.morton3d(1, 0, 0) == 613566756 and oracle.morton3d(0, 0, 1) == 153391689
    # bit interleave definition: x -> bits 3k+2, y -> 3k+1, z -> 3k
    for x, y, z in [(1, 2, 3), (1023, 0, 512), (77, 900, 5)]:
        code = sum(((x >> k) & 1) << (3 * k + 2) | ((y >> k) & 1) << (3 * k + 1) | ((z >> k) & 1) << (3 * k) for k in range(10))
        assert oracle.morton3d((x + 0.5) / 1024, (y + 0.5) / 1024, (z + 0.5) / 1024) == code


def test_neighbor_offsets(oracle):
    no = oracle.neighbor_offsets(8192)
    assert np.array_equal((no[:6] * 127).round().astype(int), [[-62, -109], [67, -73], [4, 70], [-57, -38], [72, -2], [9, -112]])
    k = no * 127
    assert np.all(np.abs(k) <= 127) and np.all(k == np.round(k)) and np.all((k ** 2).sum(1) <= 127.0 ** 2 + 1e-3)


def test_hashgrid_layout(oracle):
    total, off, res, sc = oracle.hashgrid_layout()
    assert total == 6299960 and total * 2 == 12599920
    assert res.tolist() == [16, 24, 34, 49, 71, 102, 148, 213, 308, 446, 646, 934, 1352, 1956, 2831, 4096]
    assert np.diff(off.astype(np.int64)).tolist() == [4096, 13824, 39304, 117656, 357912] + [524288] * 11
    np.testing.assert_allclose(sc[1] + 1, 16 * 1.447269237, rtol=1e-6)


def test_fp16_conversion_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(size=20000).astype(np.float32) * 10.0 ** rng.integers(-9, 5, 20000), np.array([0, -0.0, 65504, 65520, 1e-8, 6e-8, 5.96e-8, 2.98e-8, 3e-8, np.inf, -np.inf], np.float32)]).astype(np.float32)
    got = oracle.to_f16_bits(x)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, ref)
    back = np.array([oracle.lib().orc_f16_to_f32(int(h)) for h in got[:2000]], np.float32)
    assert np.array_equal(back, ref[:2000].view(np.float16).astype(np.float32))


def test_oct_roundtrip(oracle):
    rng = np.random.default_rng(2)
    for _ in range(300):
        n = rng.normal(size=3).astype(np.float32); n /= np.linalg.norm(n)
        e = oracle.oct_encode(n)
        assert 0 <= e[0] <= 1 and 0 <= e[1] <= 1
        np.testing.assert_allclose(oracle.oct_decode(e), n, atol=3e-6)
    np.testing.assert_allclose(oracle.oct_decode(np.zeros(2, np.float32)), [0, 0, -1], atol=1e-7)   # what an invalid (0,0) tile sample decodes to


def test_env_lookup_vs_numpy(oracle):
    rng = np.random.default_rng(3)
    H, W = 8, 16
    tex = rng.random((H * W, 3)).astype(np.float32)
    d = rng.normal(size=(500, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    got = oracle.env_le(tex, W, H, d)
    th = np.arccos(d[:, 1].astype(np.float64)); ph = np.arctan2(d[:, 2].astype(np.float64), d[:, 0].astype(np.float64)); ph = np.where(ph < 0, ph + 2 * np.pi, ph)
    x = ph / (2 * np.pi) * W - 0.5; y = (1 - th / np.pi) * H - 0.5
    x0 = np.trunc(x).astype(int); y0 = np.trunc(y).astype(int)
    cx0, cx1 = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1); cy0, cy1 = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    u = (x - cx0)[:, None]; v = (y - cy0)[:, None]
    T = tex.reshape(H, W, 3).astype(np.float64)
    ref = (T[cy0, cx0] * (1 - u) + T[cy0, cx1] * u) * (1 - v) + (T[cy1, cx0] * (1 - u) + T[cy1, cx1] * u) * v
    np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5)
    assert np.all(oracle.env_le(tex, W, H, np.array([[0, 1, 0], [0, -1, 0]], np.float32)) == 0)     # poles return 0 (lightDi.slang:125-126)


def test_normal_ao_known_answers(oracle):
    """process_normal_ao (EAWDenoise.slang:591-651) by hand: uniform normals -> 0; a vertical crease at x = 6 of a 12-wide frame -> the window
    (x offsets -4 .. 3, so it reaches 3 pixels right and 4 left) flags x = 3 .. 9: one foreign column out of seven or eight gives > 6, clamped to 1;
    a 30-degree crease gives 50 (1 - cos 30) k / n for k foreign columns of n valid ones; background pixels are 0 and are skipped as neighbours; the value is splat
    to three channels."""
    fx, fy = 12, 10
    occ = np.ones(fx * fy, np.float32)
    flat = np.tile(np.array([0, 0, 1], np.float32), (fx * fy, 1))
    assert np.all(oracle.normal_ao(fx, fy, occ, flat) == 0)
    crease = flat.copy().reshape(fy, fx, 3); crease[:, 6:] = [1, 0, 0]
    a = oracle.normal_ao(fx, fy, occ, crease.reshape(-1, 3)).reshape(fy, fx, 3)
    assert np.array_equal(a[..., 0], a[..., 1]) and np.array_equal(a[..., 0], a[..., 2])
    assert a[5, :, 0].tolist() == [0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0]
    c30 = np.float32(np.cos(np.deg2rad(30.0))); s30 = np.float32(np.sin(np.deg2rad(30.0)))
    soft = flat.copy().reshape(fy, fx, 3); soft[:, 6:] = [s30, 0, c30]
    b = oracle.normal_ao(fx, fy, occ, soft.reshape(-1, 3)).reshape(fy, fx, 3)[5, :, 0]
    d = float(np.float32(s30 * 0 + 0 + c30 * 1))
    assert abs(b[3] - 50 * (1 - (6 + d) / 7)) < 1e-5        # x = 3 sees columns 0 .. 6 (x - 4 is outside): one foreign of seven
    assert b[5] == 1.0 and b[2] == 0        # three foreign columns of eight: 50 (1 - cos 30) 3 / 8 = 2.5, clamped
    # x = 9 sees columns 5 .. 11 only (7 valid): one foreign
    assert abs(b[9] - 50 * (1 - (6 + d) / 7)) < 1e-5
    hole = occ.copy().reshape(fy, fx); hole[:, 6:] = 0
    h = oracle.normal_ao(fx, fy, hole.reshape(-1), crease.reshape(-1, 3)).reshape(fy, fx, 3)
    assert np.all(h[:, 6:] == 0) and np.all(h[:, :6] == 0)          # the foreign half is background: not counted, and itself 0
