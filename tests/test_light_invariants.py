"""CPU: what the environment-light sampler must satisfy whatever its code looks like (SURVEY §8 rows a-11 ... a-13; make_sampleable.slang:34-86,
lightDi.slang:67-86,181-209,285-330, GenerateLightTiles.slang) — written from the published construction (a piecewise-constant 2-D distribution over a
lat-long map, luminance x sin(theta), sampled by inverting the marginal and the conditional CDF; Pharr et al., Distribution2D) in float64 numpy, not by calling
the oracle's sampler on itself: the tables are distributions; the texels the light tiles draw follow them (chi-square); the solid-angle density stored with
every tile sample is the table density at the direction it decodes to, over 2 pi^2 sin(theta); and the estimator the whole direct-lighting chain rests on,
mean(Le / p) over the tile samples, equals a quadrature of the very function the renderer shades with (env_le) over the sphere."""
import numpy as np
import pytest

LUM = np.array([0.212671, 0.715160, 0.072169])


def oct_decode(f):
    """The published octahedral decode (Cigolle et al. 2014; helperDi.slang:122-134), batched, float64."""
    f = 2 * np.asarray(f, np.float64) - 1
    n = np.stack([f[:, 0], f[:, 1], 1 - np.abs(f[:, 0]) - np.abs(f[:, 1])], 1)
    t = np.clip(-n[:, 2], 0, 1)
    n[:, 0] += np.where(n[:, 0] >= 0, -t, t); n[:, 1] += np.where(n[:, 1] >= 0, -t, t)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def lookup_dir(d):
    """ngp_dir (lightDi.slang:432-436): the direction a WORLD direction is looked up with in the lat-long map."""
    return np.stack([-d[:, 0], d[:, 2], d[:, 1]], 1)


@pytest.fixture(scope="module")
def light(oracle, scene_mod):
    from util import SmallFrame
    F = SmallFrame(oracle, scene_mod, fx=8, fy=8, subdiv=2, ground=4, env_hw=(16, 32))
    pdf, cdf, mpdf, mcdf = (np.asarray(t, np.float64) for t in F.tables)
    H, W = F.Hc, F.Wc
    tiles = [oracle.light_tiles(F.frame, fi) for fi in (3, 1003, 7, 11)]
    # the reference masks seed coordinates to 16 bits and seeds tile sample i with (i, i): tiles 64-127 repeat tiles 0-63 exactly. The
    # statistics below take the 65 536 independent samples of each frame index; the repetition itself is asserted here, as the quirk it is
    for t in tiles:
        assert np.array_equal(t[0][:65536], t[0][65536:]) and np.array_equal(t[1][:65536], t[1][65536:]) and np.array_equal(t[2][:65536], t[2][65536:])
    tiles = [tuple(x[:65536] for x in t) for t in tiles]
    ld = np.concatenate([t[0] for t in tiles]); uv = np.concatenate([t[1] for t in tiles]); p = np.concatenate([t[2] for t in tiles]).astype(np.float64)
    # light_uv are TEXEL coordinates of the (flipped) texture the renderer looks up — v = 1 - theta / pi (lightDi.slang:119-132), so texel row y holds the
    # table's row Hc - 1 - y (the tables are indexed by theta); columns coincide
    uv = np.stack([uv[:, 0], H - 1 - uv[:, 1]], 1)
    return dict(F=F, pdf=pdf.reshape(H, W), cdf=cdf.reshape(H, W + 1), mpdf=mpdf, mcdf=mcdf, H=H, W=W, ld=ld, uv=uv, p=p)


def test_the_tables_are_distributions(light):
    pdf, cdf, mpdf, mcdf, H, W = (light[k] for k in ("pdf", "cdf", "mpdf", "mcdf", "H", "W"))
    assert (pdf >= 0).all() and (mpdf >= 0).all()
    np.testing.assert_allclose(pdf.sum(1), 1.0, rtol=0, atol=2e-5)                 # every row's conditional distribution
    np.testing.assert_allclose(mpdf.sum(), 1.0, rtol=0, atol=2e-5)                  # the marginal over rows
    assert (np.diff(cdf, axis=1) >= -1e-7).all() and np.abs(cdf[:, 0]).max() == 0 and np.abs(cdf[:, -1] - 1).max() == 0
    assert (np.diff(mcdf) >= -1e-7).all() and mcdf[0] == 0 and mcdf[-1] == 1
    np.testing.assert_allclose(np.diff(cdf, axis=1), pdf, rtol=0, atol=3e-6)        # the CDFs are the running sums of the pdfs
    np.testing.assert_allclose(np.diff(mcdf), mpdf, rtol=0, atol=3e-6)
    # the joint density is luminance x sin(theta) of the map the renderer looks up: texel centres through env_le, the function shading uses
    F = light["F"]
    th = np.pi * (np.arange(H) + 0.5) / H; ph = 2 * np.pi * (np.arange(W) + 0.5) / W
    d_world = np.stack([np.sin(th)[:, None] * np.cos(ph)[None], np.cos(th)[:, None] * np.ones(W)[None], np.sin(th)[:, None] * np.sin(ph)[None]], -1).reshape(-1, 3)
    le = F.O.env_le(F.tex, W, H, lookup_dir(d_world).astype(np.float32)).astype(np.float64)    # O.env_le takes the map-frame direction, as the Slang function does
    w = (le @ LUM).reshape(H, W) * np.sin(th)[:, None]
    np.testing.assert_allclose(pdf * mpdf[:, None], w / w.sum(), rtol=2e-4, atol=1e-9)


def test_tile_samples_follow_the_tables(light):
    from scipy import stats
    pdf, mpdf, H, W, uv = (light[k] for k in ("pdf", "mpdf", "H", "W", "uv"))
    n = len(uv)
    assert uv[:, 0].min() >= 0 and uv[:, 0].max() < W and uv[:, 1].min() >= 0 and uv[:, 1].max() < H
    rows = np.bincount(uv[:, 1], minlength=H).astype(np.float64)
    keep = n * mpdf >= 20                                                            # rows near the poles carry next to nothing: pooled
    obs = np.append(rows[keep], rows[~keep].sum()); exp = np.append(n * mpdf[keep], n * mpdf[~keep].sum())
    if exp[-1] < 5: obs, exp = obs[:-1], exp[:-1] * (obs[:-1].sum() / exp[:-1].sum())
    assert stats.chisquare(obs, exp * obs.sum() / exp.sum()).pvalue > 1e-4
    for h in np.argsort(-mpdf)[:4]:                                                  # the conditional distribution within the four heaviest rows
        sel = uv[uv[:, 1] == h, 0]
        cols = np.bincount(sel, minlength=W).astype(np.float64); e = len(sel) * pdf[h]
        k = e >= 10
        o2 = np.append(cols[k], cols[~k].sum()); e2 = np.append(e[k], e[~k].sum())
        if e2[-1] < 5: o2, e2 = o2[:-1], e2[:-1] * (o2[:-1].sum() / e2[:-1].sum())
        assert stats.chisquare(o2, e2 * o2.sum() / e2.sum()).pvalue > 1e-4, h


def test_stored_density_is_the_table_density_per_solid_angle(light):
    O = light["F"].O
    pdf, mpdf, H, W, ld, uv, p = (light[k] for k in ("pdf", "mpdf", "H", "W", "ld", "uv", "p"))
    ok = ld[:, 0] > 0.5
    assert ok.mean() > 0.999 and (p[~ok] == 0).all()                                # a sample on a pole (|sin theta| < 1e-4) is invalid and carries density 0
    d = oct_decode(ld[ok, 1:3])                                                      # the direction later passes will see
    # the tables chart WORLD directions with the pole along y — (sin t cos p, cos t, sin t sin p), lightDi.slang:199-201, 312-330 — and weigh texel (h, w) by
    # the radiance seen along that world direction, env_le(ngp_dir(.)) (make_sampleable.slang:34-60; checked in test_the_tables_are_distributions)
    th = np.arccos(np.clip(d[:, 1], -1, 1)); ph = np.arctan2(d[:, 2], d[:, 0]); ph = np.where(ph < 0, ph + 2 * np.pi, ph)
    row = np.clip((th / np.pi * H).astype(int), 0, H - 1); col = np.clip((ph / (2 * np.pi) * W).astype(int), 0, W - 1)
    same = (row == uv[ok, 1]) & (col == uv[ok, 0])
    assert same.mean() > 0.995                                                       # (the octahedral code moves a direction by ~1e-4: a few samples cross a texel border)
    want = pdf[uv[ok, 1], uv[ok, 0]] * mpdf[uv[ok, 1]] * W * H / (2 * np.pi ** 2 * np.sin(th))
    s = same & (np.sin(th) > 0.05)
    np.testing.assert_allclose(p[ok][s], want[s], rtol=3e-3)


def test_light_sampling_estimates_the_integral_of_what_the_renderer_looks_up(light):
    O, F = light["F"].O, light["F"]
    H, W, ld, p = (light[k] for k in ("H", "W", "ld", "p"))
    ok = (ld[:, 0] > 0.5) & (p > 0)
    d = oct_decode(ld[ok, 1:3])
    le = O.env_le(F.tex, W, H, lookup_dir(d).astype(np.float32)).astype(np.float64)          # what shading sees along the sampled world direction
    est = (le / p[ok, None]).sum(0) / len(p)                                          # invalid samples count as zeros
    # quadrature of env_le over the sphere on a grid 8 x finer than the map (mid-point rule in (theta, phi))
    S = 8; Hq, Wq = H * S, W * S
    th = np.pi * (np.arange(Hq) + 0.5) / Hq; ph = 2 * np.pi * (np.arange(Wq) + 0.5) / Wq
    dw = np.stack([np.sin(th)[:, None] * np.cos(ph)[None], np.cos(th)[:, None] * np.ones(Wq)[None], np.sin(th)[:, None] * np.sin(ph)[None]], -1).reshape(-1, 3)
    lq = O.env_le(F.tex, W, H, lookup_dir(dw).astype(np.float32)).astype(np.float64).reshape(Hq, Wq, 3)      # world directions on a (theta, phi) grid
    ref = (lq * np.sin(th)[:, None, None]).sum((0, 1)) * (np.pi / Hq) * (2 * np.pi / Wq)
    np.testing.assert_allclose(est @ LUM, ref @ LUM, rtol=0.01)                      # the luminance the density follows: nearly zero variance
    np.testing.assert_allclose(est, ref, rtol=0.03)                                  # and every channel on its own
