"""CPU: the oracle's BRDF libraries, FinalShading and the indirect (path-traced) half held to statements that do NOT come from the oracle's own code
(VERDICT r4 "What's missing" 1): the published formulas written out here in float64 numpy —

    GGX normal distribution           D(h) = a^2 / (pi ((n.h)^2 (a^2 - 1) + 1)^2)                       Walter et al. 2007, eq. 33
    Smith masking                     Lambda(w) = (-1 + sqrt(1 + a^2 tan^2 theta)) / 2                  Heitz 2014, eq. 72
                                      G2 = 1 / (1 + Lambda(wo) + Lambda(wi))  (height-correlated)       Heitz 2014, eq. 99;  separable: G1(wo) G1(wi)
    Fresnel                           F = F0 + (1 - F0) (1 - cos)^5                                     Schlick 1994
    microfacet BRDF                   f = F D G / (4 (n.wo) (n.wi))                                     Walter et al. 2007, eq. 20
    half-vector sampling pdf          p(wi) = D(h) (n.h) / (4 (wo.h))                                   Walter et al. 2007, eq. 24 / 38
    multiple importance sampling      w_a = p_a^2 / (p_a^2 + p_b^2)                                     Veach 1997, power heuristic, beta = 2
    rendering equation                L_o = int f L_i cos                                               Kajiya 1986

— and properties any correct implementation has (a pdf integrates to the probability of producing a sample, samples are distributed as the pdf says, a lobe reflects at
most the light it receives, a furnace never returns more than it holds, the MIS estimator of the bounce kernels converges to the integral a brute-force estimator with
uniform hemisphere sampling and no MIS converges to).  The oracle is only ever the thing under test here; geometry queries (closest hit / occlusion) go through
oracle.trace, which tests/test_oracle_invariants.py pins against brute-force Moller-Trumbore.

Reference lines checked: utils/brdfDi.slang:24-84,102-135,138-328 (sh:: library), utils/brdf.slang:155-211 (rt:: library), FinalShading.slang:11-111 (FinalShading),
:113-265 (process_new_dir_for_pt), :641-1009 (process_path_tracing_divided_no_grad), helperDi.slang:18-40 (frame), :404-409 (power heuristic), lightDi.slang:119-133 (env lookup)."""
import numpy as np
import pytest

PI = np.pi
F0_DIELECTRIC = 0.04          # FinalShading.slang:9


# ------------------------------------------------------------------------------------------------ published formulas, float64
def ggx_D(a, ch):
    a2 = a * a
    return a2 / (PI * ((ch * ch) * (a2 - 1.0) + 1.0) ** 2)


def smith_lambda(a, c):
    c = np.clip(c, 1e-12, 1.0)
    t2 = (1.0 - c * c) / (c * c)
    return 0.5 * (-1.0 + np.sqrt(1.0 + a * a * t2))


def G_correlated(a, ci, co):
    return 1.0 / (1.0 + smith_lambda(a, ci) + smith_lambda(a, co))


def G_separable(a, ci, co):
    return 1.0 / ((1.0 + smith_lambda(a, ci)) * (1.0 + smith_lambda(a, co)))


def schlick(f0, c):
    return f0 + (1.0 - f0) * np.clip(1.0 - c, 0.0, None) ** 5


def lum(v):
    return v[..., 0] * 0.212671 + v[..., 1] * 0.715160 + v[..., 2] * 0.072169          # Rec. 709 (helper.slang:101)


def nrm(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def spec_f_cos(wo, wi, f0, a, correlated=True):
    """Microfacet specular term times cos(theta_i), local frame (z = n): F D G / (4 n.wo).  f0 [.., 3] or scalar."""
    h = nrm(wo + wi)
    woh = (wo * h).sum(-1)
    G = (G_correlated if correlated else G_separable)(a, wo[..., 2], wi[..., 2])
    F = schlick(np.asarray(f0, np.float64), woh[..., None] if np.ndim(f0) else woh)
    DG = ggx_D(a, h[..., 2]) * G / (4.0 * wo[..., 2])
    ok = np.minimum(wo[..., 2], wi[..., 2]) >= 1e-6
    return np.where(ok[..., None], F * DG[..., None], 0.0) if np.ndim(f0) else np.where(ok, F * DG, 0.0)


def spec_pdf(wo, wi, a):
    h = nrm(wo + wi)
    woh = (wo * h).sum(-1)
    ok = np.minimum(wo[..., 2], wi[..., 2]) >= 1e-6
    return np.where(ok, ggx_D(a, h[..., 2]) * h[..., 2] / (4.0 * np.where(ok, woh, 1.0)), 0.0)


def diff_pdf(wo, wi):
    return np.where(np.minimum(wo[..., 2], wi[..., 2]) >= 1e-6, wi[..., 2] / PI, 0.0)


def falcor_lobes(kd, rough, metal, n_dot_v):
    """Falcor's lobe selection as the reference uses it (no transmission): pD ~ lum(kd) (1 - m), pS ~ lum(F(F0, n.v)), normalised; a = rough^2 (0 below 1e-4)."""
    kd = np.asarray(kd, np.float64)
    f0 = F0_DIELECTRIC * (1.0 - metal)[..., None] + kd * metal[..., None]
    pD = lum(kd) * (1.0 - metal)
    pS = lum(schlick(f0, n_dot_v[..., None])) * (metal + (1.0 - metal))
    s = pD + pS
    s = np.where(s > 0, s, 1.0)
    a = rough * rough
    return pD / s, pS / s, np.where(a < 1e-4, 0.0, a), f0


def hemisphere_grid(n_mu, n_phi):
    """Mid-point rule over the upper hemisphere in (mu = cos theta, phi): directions [n_mu * n_phi, 3] and the solid angle of one cell."""
    mu = (np.arange(n_mu) + 0.5) / n_mu
    ph = (np.arange(n_phi) + 0.5) * (2 * PI / n_phi)
    M, P = np.meshgrid(mu, ph, indexing="ij")
    r = np.sqrt(1 - M * M)
    return np.stack([r * np.cos(P), r * np.sin(P), M], -1).reshape(-1, 3), (1.0 / n_mu) * (2 * PI / n_phi)


def view_dir(nv):
    return np.array([np.sqrt(1 - nv * nv), 0.0, nv])


def random_states(n, seed):
    """Generator states for the oracle's LCG: independent 32-bit words (consecutive integers would give correlated first draws)."""
    return np.random.default_rng(seed).integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)


def chi2_check(counts, expected, what):
    """Pearson chi-square with cells of expectation < 8 pooled; fails beyond mean + 5 sigma of the chi-square distribution."""
    big = expected >= 8
    c = np.append(counts[big], counts[~big].sum()); e = np.append(expected[big], expected[~big].sum())
    keep = e > 0
    stat = float((((c - e) ** 2)[keep] / e[keep]).sum()); dof = int(keep.sum()) - 1
    assert dof > 50, (what, dof)
    assert stat < dof + 5.0 * np.sqrt(2.0 * dof), "%s: chi2 %.1f for %d degrees of freedom" % (what, stat, dof)
    return stat, dof


def bin_directions(w, n_mu, n_phi):
    mu = np.clip((w[:, 2] * n_mu).astype(int), 0, n_mu - 1)
    ph = np.arctan2(w[:, 1], w[:, 0]); ph = np.where(ph < 0, ph + 2 * PI, ph)
    return np.bincount(mu * n_phi + np.clip((ph / (2 * PI) * n_phi).astype(int), 0, n_phi - 1), minlength=n_mu * n_phi).astype(np.float64)


def bin_integrals(pdf_fn, n_mu, n_phi, sub=12):
    """Integral of pdf_fn over every (mu, phi) cell of an n_mu x n_phi partition of the hemisphere (mid-point rule on a sub x sub refinement)."""
    w, dw = hemisphere_grid(n_mu * sub, n_phi * sub)
    p = pdf_fn(w).reshape(n_mu, sub, n_phi, sub)
    return p.sum(axis=(1, 3)).reshape(-1) * dw


# ------------------------------------------------------------------------------------------------ (c) point values against the formulas
def test_oracle_brdf_point_values_equal_the_published_formulas(oracle):
    """10^4 random (wo, wi, roughness, metallic, albedo): SpecularReflection_eval / _evalPdf, Diffuse_light, FalcorBRDF_eval / _evalPdf, the lobe probabilities of
    FinalShading.slang:58-78, and brdf.slang's evalBRDF / evalPdfBRDF equal the float64 formulas.  Tolerance: a few float32 roundings, amplified where GGX's denominator
    ((n.h)^2 (a^2 - 1) + 1) cancels (relative error of that subtraction = 6e-8 / its value)."""
    rng = np.random.default_rng(11)
    n = 10000
    def hemi(k):
        v = rng.normal(size=(k, 3)); v[:, 2] = np.abs(v[:, 2]) + 0.02
        return nrm(v)
    wo, wi = hemi(n), hemi(n)
    rough = rng.uniform(0.05, 1.0, n); metal = rng.uniform(0, 1, n) * (rng.random(n) < 0.7); kd = rng.uniform(0.05, 1.0, (n, 3))
    normal = nrm(rng.normal(size=(n, 3)))
    # a view direction with the prescribed n.v = wo.z around `normal` (any azimuth)
    t = nrm(np.cross(normal, rng.normal(size=(n, 3))))
    view = normal * wo[:, 2:3] + t * np.sqrt(1 - wo[:, 2:3] ** 2)
    pD, pS, a, f0 = falcor_lobes(kd, rough, metal, wo[:, 2])
    opD, opS, oa, of0 = oracle.sh_lobes(kd, rough, metal, -view, normal)
    np.testing.assert_allclose(opD, pD, rtol=2e-5, atol=1e-7); np.testing.assert_allclose(opS, pS, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(oa, a, rtol=1e-6); np.testing.assert_allclose(of0, f0, rtol=1e-6, atol=1e-8)
    diff_albedo = kd * (1 - metal)[:, None]
    o = oracle.sh_eval(pD, pS, a, f0, diff_albedo, wo, wi)
    h = nrm(wo + wi)
    d = (h[:, 2] ** 2) * (a * a - 1) + 1
    tol = 4e-6 * (1 + 1 / d)                                     # two appearances of d (squared) + ~20 other roundings
    sf = spec_f_cos(wo, wi, f0, a); sp = spec_pdf(wo, wi, a)
    assert np.all(np.abs(o["spec_f"] - sf) <= tol[:, None] * np.abs(sf) + 1e-30), float(np.max(np.abs(o["spec_f"] - sf) / (np.abs(sf) + 1e-30) / tol[:, None]))
    assert np.all(np.abs(o["spec_pdf"] - sp) <= tol * sp + 1e-30)
    np.testing.assert_allclose(o["diff_light"], diff_pdf(wo, wi), rtol=2e-6)
    f = diff_albedo * diff_pdf(wo, wi)[:, None] + sf
    p = pD * diff_pdf(wo, wi) + pS * sp
    assert np.all(np.abs(o["f"] - f) <= tol[:, None] * np.abs(f) + 1e-30) and np.all(np.abs(o["pdf"] - p) <= tol * p + 1e-30)
    assert sf.max() > 1.0 and sp.max() > 1.0                      # the sample reaches into the lobes' peaks
    # brdf.slang (reservoir target / candidate pdf): separable Smith, scalar Fresnel divided by its F0, lerp(specular, diffuse, mix), saturate()d cosines
    wd = lum(kd); ws = lum(np.repeat(metal[:, None], 3, 1)); al = np.clip(rough, 0.01, 1.0) ** 2          # brdf_map of restir_di_with_pt (renderer_restir.py:279-287)
    fr, pr = oracle.rt_eval(wi, wo, np.array([0, 0, 1.0]), al, wd, ws)           # local frame as world frame: N = z
    mix = np.where(wd + ws > 1e-7, wd / np.where(wd + ws > 1e-7, wd + ws, 1), 1.0)
    hd = (h * wi).sum(1)
    Fs = np.where(ws < 1e-8, 0.0, schlick(ws, hd) / np.where(ws < 1e-8, 1.0, ws))
    spec = np.maximum(0.0, ggx_D(al, h[:, 2]) * G_separable(al, wo[:, 2], wi[:, 2]) * Fs / (4 * wo[:, 2]))
    want = spec * (1 - mix) + (wi[:, 2] / PI) * mix
    d2 = (h[:, 2] ** 2) * (al * al - 1) + 1
    tol2 = 4e-6 * (1 + 1 / d2)
    assert np.all(np.abs(fr - want) <= tol2 * want + 1e-30)
    wantp = (ggx_D(al, h[:, 2]) * h[:, 2] / (4 * (h * wo).sum(1))) * (1 - mix) + (wi[:, 2] / PI) * mix
    assert np.all(np.abs(pr - wantp) <= 4 * tol2 * wantp + 1e-30)


def test_power_heuristic_and_frame(oracle):
    """create_frame is an orthonormal right-handed basis around the normal for every normal (Duff et al. 2017's construction; helperDi.slang:18-30)."""
    rng = np.random.default_rng(3)
    for nvec in list(nrm(rng.normal(size=(200, 3)))) + [np.array([0, 0, 1.0]), np.array([0, 0, -1.0]), np.array([1.0, 0, 0]), nrm(np.array([1e-4, 0, -1.0]))]:
        x, y = oracle.sh_frame(nvec)
        B = np.stack([x, y, nvec.astype(np.float32)]).astype(np.float64)
        np.testing.assert_allclose(B @ B.T, np.eye(3), atol=3e-6)
        assert np.linalg.det(B) > 0.999


# ------------------------------------------------------------------------------------------------ (a) normalisation and energy
GRID = [(r, m, nv) for r in (0.16, 0.4, 1.0) for m in (0.0, 0.5, 1.0) for nv in (0.2, 0.6, 0.95)]


def test_pdfs_integrate_to_the_probability_of_a_sample_and_lobes_conserve_energy(oracle):
    """For a grid of (roughness, metallic, n.v): the oracle's FalcorBRDF_evalPdf integrates over the hemisphere to pD + pS (1 - loss), loss = the part of the half-vector
    distribution whose reflection falls below the horizon — measured as the fraction of invalid draws of the oracle's own sampler, and computed from the published
    pdf on the same quadrature; never above 1.  The specular lobe with F = 1 reflects at most what it receives (int f cos <= 1), the diffuse lobe exactly its albedo,
    brdf.slang's evalBRDF (a convex mix of a diffuse lobe and a specular one whose Fresnel term is normalised by F0) at most mix + (1 - mix) / F0, and its pdf
    integrates like the Falcor one."""
    w, dw = hemisphere_grid(1100, 2200)
    kd = np.array([0.7, 0.5, 0.3])
    for rough, metal, nv in GRID:
        wo = view_dir(nv)
        pD, pS, a, f0 = falcor_lobes(kd[None], np.array([rough]), np.array([metal]), np.array([nv]))
        pD, pS, a, f0 = float(pD[0]), float(pS[0]), float(a[0]), f0[0]
        o = oracle.sh_eval(pD, pS, a, [1, 1, 1], [1, 1, 1], np.repeat(wo[None], len(w), 0), w)
        total = float(o["pdf"].astype(np.float64).sum() * dw)
        published = float((pD * diff_pdf(wo[None], w) + pS * spec_pdf(wo[None], w, a)).sum() * dw)
        assert abs(total - published) < 2e-3, (rough, metal, nv, total, published)
        assert total <= 1.0 + 2e-3
        s = oracle.sh_sample(random_states(200000, 5), pD, pS, a, f0, kd * (1 - metal), wo, True)
        p_valid = float(s["valid"].mean())
        assert abs(total - p_valid) < 6e-3, (rough, metal, nv, total, p_valid)
        # energy: specular lobe with white Fresnel, diffuse lobe with albedo 1
        e_spec = float(o["spec_f"][:, 0].astype(np.float64).sum() * dw)
        assert e_spec <= 1.0 + 2e-3, (rough, nv, e_spec)
        assert e_spec > 0.25                                         # ... and it is a lobe, not a hole (single scattering loses energy at high roughness / grazing views, not everything)
        e_diff = float(o["diff_light"].astype(np.float64).sum() * dw)
        assert abs(e_diff - 1.0) < 1e-3
        # brdf.slang
        wd, ws = float(lum(kd)), float(metal)
        al = max(rough, 0.01) ** 2
        fr, pr = oracle.rt_eval(w, wo, np.array([0, 0, 1.0]), al, wd, ws)
        mix = wd / (wd + ws)
        # the target function is not a physical BRDF: its Fresnel term is divided by F0 (1 at normal incidence, up to 1 / F0 at grazing angles), so the specular part
        # may return up to 1 / ws of what it receives; the convex mix bounds the whole
        assert float(fr.astype(np.float64).sum() * dw) <= mix + (1 - mix) * (1.0 / ws if ws > 0 else 0.0) + 2e-3
        h = nrm(wo[None] + w)
        pub = float(((ggx_D(al, h[:, 2]) * h[:, 2] / (4 * (h * wo[None]).sum(1))) * (1 - mix) + w[:, 2] / PI * mix).sum() * dw)
        tot_rt = float(pr.astype(np.float64).sum() * dw)
        assert abs(tot_rt - pub) < 2e-3 and tot_rt <= 1.0 + 2e-3, (rough, metal, nv, tot_rt, pub)


# ------------------------------------------------------------------------------------------------ (b) the samplers draw from the pdfs they report
@pytest.mark.parametrize("rough,metal,nv", [(0.6, 0.0, 0.7), (0.3, 0.8, 0.4), (0.2, 0.3, 0.9), (1.0, 0.0, 0.25)])
def test_falcor_sample_is_distributed_as_the_published_pdf(oracle, rough, metal, nv):
    """chi-square of 4e5 draws of FalcorBRDF_sample over a 16 x 32 partition of the hemisphere against pD cos/pi + pS D (n.h) / (4 wo.h) (float64); the pdf the sampler
    REPORTS equals that mixture for every valid draw (both branches: roughness > 0.15), weight = f / pdf, and exactly four numbers are drawn (one to select, one burnt, two)."""
    kd = np.array([0.8, 0.6, 0.4])
    wo = view_dir(nv)
    pD, pS, a, f0 = falcor_lobes(kd[None], np.array([rough]), np.array([metal]), np.array([nv]))
    pD, pS, a, f0 = float(pD[0]), float(pS[0]), float(a[0]), f0[0]
    diff_albedo = kd * (1 - metal)
    n = 400000
    sg = random_states(n, 21)
    s = oracle.sh_sample(sg, pD, pS, a, f0, diff_albedo, wo, True)
    v = s["valid"]
    wi = s["wi"][v].astype(np.float64)
    mix = lambda w: pD * diff_pdf(wo[None], w) + pS * spec_pdf(wo[None], w, a)
    cell = bin_integrals(mix, 16, 32)
    chi2_check(bin_directions(wi, 16, 32), v.sum() * cell / cell.sum(), "FalcorBRDF_sample(%g, %g, %g)" % (rough, metal, nv))
    p = mix(nrm(wi))
    hz = nrm(wo[None] + nrm(wi))[:, 2]
    tol = 2e-5 + 1e-6 / (hz * hz * (a * a - 1) + 1)          # float32 roundings of n.h, amplified where GGX's denominator cancels (the lobe's peak)
    assert np.all(np.abs(s["pdf"][v] - p) <= tol * p), float(np.max(np.abs(s["pdf"][v] - p) / p / tol))
    f = diff_albedo[None] * diff_pdf(wo[None], wi)[:, None] + spec_f_cos(np.repeat(wo[None], len(wi), 0), nrm(wi), f0, a)
    assert np.all(np.abs(s["weight"][v] - f / p[:, None]) <= 2 * tol[:, None] * (f / p[:, None]) + 1e-7)
    assert not s["specular_bounce"].any()
    # the lobe is chosen by the first number: u < pD -> diffuse
    assert abs(float((s["u_select"] < pD).mean()) - pD) < 4e-3
    st = sg.copy()
    for _ in range(4):
        st = (st.astype(np.uint64) * 1664525 + 1013904223).astype(np.uint32)          # lcg (random.slang:39-44)
    assert np.array_equal(s["sg_out"][v], st[v])


def test_falcor_sample_below_the_specular_roughness_threshold(oracle):
    """sqrt(alpha) <= 0.15 (brdfDi.slang:316-324): draws of the specular branch are flagged `specularBounce` and report pS * p_spec only, draws of the diffuse branch
    still report the mixture; the directions of all valid draws together are distributed as the mixture.  alpha < 1e-4 (FinalShading.slang:66-68 sets it to 0):
    the specular branch never yields a sample, the diffuse one reports pD cos / pi."""
    kd = np.array([0.8, 0.6, 0.4]); nv = 0.7; wo = view_dir(nv)
    rough, metal = 0.12, 0.4
    pD, pS, a, f0 = falcor_lobes(kd[None], np.array([rough]), np.array([metal]), np.array([nv]))
    pD, pS, a, f0 = float(pD[0]), float(pS[0]), float(a[0]), f0[0]
    s = oracle.sh_sample(random_states(400000, 8), pD, pS, a, f0, kd * (1 - metal), wo, True)
    v = s["valid"]; spec_branch = s["u_select"] >= np.float32(pD)
    assert np.array_equal(s["specular_bounce"][v] > 0, spec_branch[v])
    wi = nrm(s["wi"].astype(np.float64))
    ps, pd = spec_pdf(wo[None], wi, a), diff_pdf(wo[None], wi)
    hz = nrm(wo[None] + wi)[:, 2]
    tol = 2e-5 + 1e-6 / (hz * hz * (a * a - 1) + 1)          # alpha = 0.0144: GGX's denominator falls to 2e-4 at the peak, so float32 n.h is worth up to 0.5 % there
    m = v & spec_branch
    assert np.all(np.abs(s["pdf"][m] - (pS * ps)[m]) <= tol[m] * (pS * ps)[m])
    m = v & ~spec_branch
    assert np.all(np.abs(s["pdf"][m] - (pD * pd + pS * ps)[m]) <= tol[m] * (pD * pd + pS * ps)[m])
    # the lobe is 1.4 degrees wide: a polar partition around the mirror direction in half-vector space would be the sharp test; here the coarse one — the mass inside
    # a 10-degree cone around the mirror direction equals the integral of the mixture over it
    mirror = np.array([-wo[0], -wo[1], wo[2]])
    inside = (wi[v] @ mirror) > np.cos(np.radians(10))
    w, dw = hemisphere_grid(2000, 4000)
    m = (w @ mirror) > np.cos(np.radians(10))
    mass = float(((pD * diff_pdf(wo[None], w[m]) + pS * spec_pdf(wo[None], w[m], a)).sum()) * dw)
    total = float(((pD * diff_pdf(wo[None], w) + pS * spec_pdf(wo[None], w, a)).sum()) * dw)
    assert abs(inside.mean() - mass / total) < 4e-3, (inside.mean(), mass / total)
    # alpha -> 0
    pD0, pS0, a0, f00 = falcor_lobes(kd[None], np.array([0.005]), np.array([metal]), np.array([nv]))
    assert a0[0] == 0.0
    s0 = oracle.sh_sample(random_states(100000, 9), float(pD0[0]), float(pS0[0]), 0.0, f00[0], kd * (1 - metal), wo, True)
    sb = s0["u_select"] >= np.float32(pD0[0])
    assert not s0["valid"][sb].any() and s0["valid"][~sb].mean() > 0.99
    w0 = nrm(s0["wi"][s0["valid"]].astype(np.float64))
    np.testing.assert_allclose(s0["pdf"][s0["valid"]], float(pD0[0]) * w0[:, 2] / PI, rtol=3e-4)
    chi2_check(bin_directions(w0, 16, 32), s0["valid"].sum() * bin_integrals(lambda w_: w_[:, 2] / PI, 16, 32), "cosine hemisphere")


def test_reservoir_brdf_sampler_is_distributed_as_its_pdf(oracle):
    """brdf.slang sampleBRDF (the BRDF candidates of the initial resampling) against evalPdfBRDF's published form, world-space normal not along an axis."""
    rng = np.random.default_rng(17)
    N = nrm(np.array([0.3, -0.5, 0.8])); t = nrm(np.cross(N, [0, 0, 1.0])); b = np.cross(N, t)
    nv = 0.6; V = N * nv + t * np.sqrt(1 - nv * nv)
    al, wd, ws = 0.3 ** 2, 0.55, 0.35
    xi = rng.random((400000, 3))
    d, ok = oracle.rt_sample(xi, V, N, al, wd, ws)
    local = lambda v: np.stack([v @ t, v @ b, v @ N], -1)
    wl = local(d[ok].astype(np.float64)); vo = local(V[None])[0]
    mix = wd / (wd + ws)
    def pdf(w):
        h = nrm(vo[None] + w)
        return mix * w[:, 2] / PI + (1 - mix) * ggx_D(al, h[:, 2]) * h[:, 2] / (4 * (h * vo[None]).sum(1))
    cell = bin_integrals(pdf, 16, 32)
    chi2_check(bin_directions(wl, 16, 32), ok.sum() * cell / cell.sum(), "sampleBRDF")
    assert abs(ok.mean() - cell.sum()) < 4e-3
    _, pr = oracle.rt_eval(d[ok], V, N, al, wd, ws)
    np.testing.assert_allclose(pr, pdf(nrm(wl)), rtol=1e-3)


# ------------------------------------------------------------------------------------------------ (c) FinalShading against the formula
def _bare_frame(oracle, fx, fy, normal, ray_dir):
    """An OrcFrame whose only meaningful members are the ones FinalShading reads (occupancy 1 everywhere, the given ray directions); one dummy triangle, a 2 x 4 map."""
    N = fx * fy
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5], [6, 5, 5], [5, 6, 5]], np.float32); t = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    info, aabb, _, _ = oracle.bvh_build(v, t)
    tex = np.ones((8, 3), np.float32)
    keep = oracle.Keep()
    nd = np.concatenate([normal, np.ones((N, 1), np.float32)], 1).astype(np.float32)
    fr = oracle.make_frame(keep, fx, fy, np.ones(N, np.float32), np.zeros((N, 3), np.float32), nd, np.zeros((N, 3), np.float32), ray_dir.astype(np.float32), (info, aabb), v, t, tex, 4, 2,
                           oracle.make_sampleable(tex, 4, 2))
    return fr, keep


def test_final_shading_equals_the_rendering_equation_integrand(oracle):
    """process_FinalShading on 10^4 random pixels (normals, view directions, materials, light directions and radiances): diffuse light = cos/pi Li, specular light =
    F D G / (4 n.v) Li with the height-correlated G and F0 = 0.04 (1 - m) + kd m, colour = kd (1 - m) diffuse + specular; zero below either horizon and for distance 0."""
    rng = np.random.default_rng(29)
    fx = fy = 100; n = fx * fy
    normal = nrm(rng.normal(size=(n, 3)))
    def around(nv_lo):
        t = nrm(np.cross(normal, rng.normal(size=(n, 3)))); c = rng.uniform(nv_lo, 1.0, (n, 1))
        return normal * c + t * np.sqrt(1 - c * c)
    view = around(0.05); light = around(-0.2)                       # a fifth of the lights from below the horizon
    kd = rng.uniform(0.05, 1.0, (n, 3)); rough = rng.uniform(0.2, 1.0, n); metal = rng.uniform(0, 1, n) * (rng.random(n) < 0.6)
    Li = rng.uniform(0.0, 4.0, (n, 3)); dist = np.where(rng.random(n) < 0.9, 1e6, 0.0)
    fr, keep = _bare_frame(oracle, fx, fy, normal.astype(np.float32), -view)
    f32 = lambda x: np.ascontiguousarray(x, np.float32)
    rm = f32(np.stack([rough, metal], 1))
    c, d, s = oracle.final_shading(fr, f32(normal), f32(kd), rm, f32(light), f32(dist), f32(Li))
    # float64 statement, in the world frame (no local frame needed: everything is a dot product with n)
    kd, rough, metal, Li, normal, view, light = (np.asarray(f32(x), np.float64) for x in (kd, rough, metal, Li, normal, view, light))
    nv, nl = (normal * view).sum(1), (normal * light).sum(1)
    h = nrm(view + light); nh, vh = (normal * h).sum(1), (view * h).sum(1)
    a = rough * rough
    f0 = F0_DIELECTRIC * (1 - metal)[:, None] + kd * metal[:, None]
    lit = (np.minimum(nv, nl) >= 1e-6) & (dist > 0)
    with np.errstate(all="ignore"):
        diff = np.where(lit, nl / PI, 0.0)[:, None] * Li
        spec = np.where(lit[:, None], schlick(f0, vh[:, None]) * (ggx_D(a, nh) * G_correlated(a, nv, nl) / (4 * nv))[:, None], 0.0) * Li
    col = kd * (1 - metal)[:, None] * diff + spec
    dden = nh * nh * (a * a - 1) + 1
    tol = (1e-5 * (1 + 0.2 / dden))[:, None]
    edge = np.abs(np.minimum(nv, nl) - 1e-6) < 1e-6                  # float32 / float64 may disagree which side of the horizon test a grazing direction is on
    for got, want in ((d, diff), (s, spec), (c, col)):
        bad = (np.abs(got - want) > tol * np.abs(want) + 1e-7) & ~edge[:, None]
        assert not bad.any(), (int(bad.sum()), float(np.max(np.abs(got - want)[bad] / (np.abs(want)[bad] + 1e-7))))
    assert lit.mean() > 0.6 and (spec > 1.0).any()


# ------------------------------------------------------------------------------------------------ scenes for (d) and (e)
def analytic_env(H, W):
    """A smooth, low-dynamic-range sky L(d) (z up) and its H x W lat-long map in the layout the reference expects: row 0 = zenith, column u = atan2(d.y, -d.x) / 2 pi
    (env_le o ngp_dir with the vertical flip of renderer_restir.py:305-311: theta = acos(d.z), phi = atan2(d.y, -d.x), v = 1 - theta / pi on the flipped map).
    Smooth enough for the map's bilinear reconstruction to be within a fraction of a percent of L at 32 x 64."""
    s = nrm(np.array([0.4, -0.5, 0.75]))
    def L(d):
        d = np.asarray(d, np.float64)
        c = np.clip(d @ s, 0, None)
        up = 0.5 + 0.5 * d[..., 2]
        return np.stack([0.35 + 0.9 * c ** 2 + 0.25 * up, 0.40 + 0.7 * c ** 2 + 0.35 * up, 0.55 + 0.5 * c ** 2 + 0.5 * up], -1)
    th = (np.arange(H) + 0.5) * PI / H; ph = (np.arange(W) + 0.5) * 2 * PI / W
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([-np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], -1)
    return L, L(d).astype(np.float32)


class Scene:
    """Icosphere on a ground plate, G-buffer by the oracle's closest-hit query (geometry only), constant material."""
    def __init__(self, O, S, fx, fy, kd, rough, metal, subdiv=2, ground=4, env_hw=(32, 64)):
        self.O, self.fx, self.fy = O, fx, fy
        self.vert, self.tri = S.make_mesh(subdiv, ground)
        self.info, self.aabb, _, _ = O.bvh_build(self.vert, self.tri)
        eye, rd = S.camera_rays(fy, fx)
        r = O.trace(self.info, self.aabb, self.vert, self.tri, O.make_rays(np.repeat(eye[None], fx * fy, 0), rd), True)
        self.occ = r["hit"].astype(np.float32); self.pos = r["pos"].copy()
        self.normal = np.where(self.occ[:, None] > 0, r["normal"], 0).astype(np.float32)
        self.depth = np.linalg.norm(self.pos - eye, axis=1).astype(np.float32)
        self.ray_dir = rd.astype(np.float32)
        N = fx * fy
        self.kd = np.tile(np.asarray(kd, np.float32), (N, 1)); self.rm = np.tile(np.array([rough, metal], np.float32), (N, 1))
        self.kd0, self.rough, self.metal = np.asarray(kd, np.float64), float(rough), float(metal)
        self.Le, self.env = analytic_env(*env_hw)

    def render(self, spp, max_bounce, seed=4242):
        O = self.O
        return O.render(self.fx, self.fy, spp, seed, (self.info, self.aabb), self.vert, self.tri, self.env, self.occ, self.normal, self.depth, self.kd, self.rm, self.ray_dir, self.pos,
                        mat=None, max_bounce=max_bounce, denoise_iter=0, want_avg=True, const_kd=tuple(self.kd0), const_rs=(self.rough, self.metal))

    # geometry queries with the reference's semantics (VIS_near offset along the direction, FinalShading.slang:8)
    def closest(self, p, d):
        r = self.O.trace(self.info, self.aabb, self.vert, self.tri, self.O.make_rays((p + 0.01 * d).astype(np.float32), d.astype(np.float32)), True)
        return r["hit"] > 0, r["pos"].astype(np.float64), r["normal"].astype(np.float64)

    def occluded(self, p, d):
        return self.O.trace(self.info, self.aabb, self.vert, self.tri, self.O.make_rays((p + 0.01 * d).astype(np.float32), d.astype(np.float32)), False)["hit"] > 0


def brdf_cos(n, v, l, kd, rough, metal):
    """(diffuse + microfacet specular) * cos(theta_l), float64, world space; n, v, l [k, 3] unit vectors."""
    nv, nl = (n * v).sum(1), (n * l).sum(1)
    ok = np.minimum(nv, nl) >= 1e-6
    h = nrm(v + l); nh, vh = (n * h).sum(1), (v * h).sum(1)
    a = rough * rough
    f0 = F0_DIELECTRIC * (1 - metal) + kd * metal
    with np.errstate(all="ignore"):
        spec = schlick(f0[None], vh[:, None]) * (ggx_D(a, nh) * G_correlated(a, nv, nl) / (4 * nv))[:, None]
        out = (kd * (1 - metal))[None] * (nl / PI)[:, None] + spec
    return np.where(ok[:, None], out, 0.0)


def uniform_hemisphere(rng, n):
    """Uniform directions on the hemisphere around n [k, 3]; pdf = 1 / (2 pi)."""
    d = nrm(rng.normal(size=n.shape))
    return np.where(((d * n).sum(1) < 0)[:, None], -d, d)


def brute_force_indirect(sc, pix, paths, bounces, rng):
    """Indirect radiance towards the camera at G-buffer pixels `pix`: sum over path vertices 1 .. bounces of throughput x direct lighting at the vertex, every
    direction (continuation and light) drawn UNIFORMLY over the hemisphere, no MIS, no next-event estimation, float64.  Returns the mean over `paths` paths per pixel."""
    k = len(pix) * paths
    p = np.repeat(sc.pos[pix].astype(np.float64), paths, 0); n = np.repeat(sc.normal[pix].astype(np.float64), paths, 0)
    rd = sc.ray_dir[pix].astype(np.float64); v = np.repeat(-nrm(rd), paths, 0)
    thr = np.ones((k, 3)); alive = np.ones(k, bool); total = np.zeros((k, 3))
    for b in range(bounces):
        w = uniform_hemisphere(rng, n)
        thr = thr * brdf_cos(n, v, w, sc.kd0, sc.rough, sc.metal) * (2 * PI)
        hit, hp, hn = sc.closest(p, w)
        alive &= hit & (thr.max(1) > 0)
        p, n, v = hp, hn, -w
        # direct lighting at the new vertex: one uniform direction, visibility by the occlusion query, radiance from the analytic sky
        l = uniform_hemisphere(rng, n)
        vis = ~sc.occluded(p, l)
        Ld = brdf_cos(n, v, l, sc.kd0, sc.rough, sc.metal) * sc.Le(l) * (2 * PI) * vis[:, None]
        total += np.where(alive[:, None], thr * Ld, 0.0)
    return total.reshape(len(pix), paths, 3).mean(1)


# ------------------------------------------------------------------------------------------------ direct lighting vs quadrature, (e) indirect estimator vs brute force
def test_direct_lighting_converges_to_a_quadrature_of_the_published_brdf(oracle, scene_mod):
    """The whole ReSTIR-DI chain (light tiles, initial / temporal / spatial resampling, final visibility, EvaluateFinalSamples, FinalShading; un-denoised) against
    sum_k f(n, v, w_k) cos L(w_k) V(w_k) dw over 4000 directions — f the float64 BRDF above (NOT the oracle's FinalShading), L the analytic sky (NOT the map lookup),
    V the oracle's occlusion query.  Colour, and the diffuse / specular light buffers separately.  Foreground means within 4 %, per-pixel correlation > 0.95."""
    sc = Scene(oracle, scene_mod, 20, 16, kd=(0.75, 0.6, 0.45), rough=0.6, metal=0.2)
    spp = 256
    out = sc.render(spp=spp, max_bounce=1)
    fg = np.flatnonzero(sc.occ > 0.5)
    K = 4000
    k = np.arange(K) + 0.5
    z = 1 - 2 * k / K; phi = k * PI * (3 - np.sqrt(5)); rr = np.sqrt(1 - z * z)
    dirs = np.stack([rr * np.cos(phi), rr * np.sin(phi), z], 1)
    n = sc.normal[fg].astype(np.float64); v = -nrm(sc.ray_dir[fg].astype(np.float64)); p = sc.pos[fg].astype(np.float64)
    col = np.zeros((len(fg), 3)); dif = np.zeros((len(fg), 3)); spe = np.zeros((len(fg), 3))
    for j in range(K):
        L = np.repeat(dirs[j][None], len(fg), 0)
        if not ((n * L).sum(1) > 1e-6).any():
            continue
        vis = (~sc.occluded(p, L))[:, None] * sc.Le(dirs[j])[None] * (4 * PI / K)
        col += brdf_cos(n, v, L, sc.kd0, sc.rough, sc.metal) * vis
        lambert = np.where((np.minimum((n * v).sum(1), (n * L).sum(1)) >= 1e-6)[:, None], ((n * L).sum(1) / PI)[:, None], 0.0)
        dif += lambert * vis                                                                                   # the diffuse LIGHT buffer carries no albedo
        spe += (brdf_cos(n, v, L, sc.kd0, sc.rough, sc.metal) - (sc.kd0 * (1 - sc.metal))[None] * lambert) * vis
    est = out["avg_direct"][fg].astype(np.float64)           # already the per-sample mean (orc_finish averages the sums in place)
    np.testing.assert_allclose(est.mean(0), col.mean(0), rtol=0.04)
    np.testing.assert_allclose(out["diffuse"][fg].astype(np.float64).mean(0), dif.mean(0), rtol=0.04)          # denoise_iter = 0: the buffers are the per-sample means
    np.testing.assert_allclose(out["spec"][fg].astype(np.float64).mean(0), spe.mean(0), rtol=0.05)
    assert np.corrcoef(lum(est), lum(col))[0, 1] > 0.95



@pytest.mark.parametrize("bounces", [2, 3])
def test_indirect_converges_to_a_brute_force_path_tracer(oracle, scene_mod, bounces):
    """The oracle's `indirect` buffer (process_new_dir_for_pt + process_path_tracing_divided_no_grad per vertex: BRDF-sampled continuation with its reported pdf,
    next-event estimation from the tabulated environment distribution and a BRDF sample, both weighted by the power heuristic) and a brute-force estimator of the same
    integral that shares nothing with it but the geometry queries: uniform hemisphere sampling everywhere, no MIS, the float64 BRDF above, the analytic sky instead of
    the map.  Foreground mean within 3 % per channel, per-pixel correlation > 0.97 (observed at 1400 spp / 5000 paths: ratios 0.989-0.991, correlation 0.9993).  Roughness 0.6 > 0.15: the reference's estimator is unbiased there
    (below, brdfDi.slang:316-324 drops the diffuse pdf from specular-branch draws, which over-counts by design)."""
    sc = Scene(oracle, scene_mod, 20, 16, kd=(0.75, 0.6, 0.45), rough=0.6, metal=0.2)
    out = sc.render(spp=1000, max_bounce=bounces)
    fg = np.flatnonzero(sc.occ > 0.5)
    est = (out["indirect_diff"] + out["indirect_spec"])[fg].astype(np.float64)
    np.testing.assert_allclose(out["indirect"][fg], est, rtol=1e-6, atol=1e-7)               # with denoise_iter = 0 the buffer is the sum of its two halves
    ref = brute_force_indirect(sc, fg, 5000, bounces, np.random.default_rng(77))
    a, b = est.mean(0), ref.mean(0)
    np.testing.assert_allclose(a, b, rtol=0.03)
    assert b.min() > 0.005                                                                    # there is indirect light to speak of (2 % of the sky's radiance here)
    cc = np.corrcoef(lum(est), lum(ref))[0, 1]
    assert cc > 0.97, cc


def test_indirect_grows_by_less_than_the_albedo_per_bounce_and_a_furnace_stays_bounded(oracle, scene_mod):
    """Uniform sky E = 1, albedo close to one (kd = 0.94, dielectric, rough: the specular lobe adds at most 0.04): every surface point receives at most E from every
    direction, so (i) the diffuse light of the direct term is at most E (not equal to E even under an open sky: triangle_hit ignores t, helperDi.slang:172-195, so a
    ray leaving a large triangle is still inside that triangle's box 0.01 further on and "hits" it at t = -0.01 — the reference's self-shadowing, kept; the direct
    term's VALUE is held to a quadrature in test_direct_lighting_converges_to_a_quadrature_of_the_published_brdf); (ii) direct + indirect radiance never exceeds E (1 + 0.04) however many bounces are added; (iii) each further bounce
    adds less than albedo x the previous one (foreground means; Monte-Carlo slack 2 %); (iv) the brute-force estimator agrees in the furnace as well."""
    sc = Scene(oracle, scene_mod, 20, 16, kd=(0.94, 0.94, 0.94), rough=1.0, metal=0.0)
    sc.env = np.ones_like(sc.env); sc.Le = lambda d: np.ones(np.shape(d)[:-1] + (3,))
    fg = sc.occ > 0.5
    ind = {}
    for mb in (1, 2, 3):
        o = sc.render(spp=256, max_bounce=mb, seed=99)
        ind[mb] = o["indirect"][fg].astype(np.float64)
        direct = o["avg_direct"][fg].astype(np.float64)
        total = direct + ind[mb]
        assert total.mean() <= 1.04 * 1.02, (mb, total.mean())
        assert np.percentile(total, 99) <= 1.04 * 1.25                       # single pixels carry Monte-Carlo noise, not energy
        dl = o["diffuse"][fg].astype(np.float64)
        assert dl.mean() <= 1.02 and dl.max() <= 1.3
    add2 = ind[2].mean() - ind[1].mean(); add3 = ind[3].mean() - ind[2].mean()
    assert 0 < add2 <= 0.98 * 1.02 * ind[1].mean(), (ind[1].mean(), add2)
    assert 0 < add3 <= 0.98 * 1.05 * add2, (add2, add3)
    ref = brute_force_indirect(sc, np.flatnonzero(fg), 1500, 2, np.random.default_rng(5))
    np.testing.assert_allclose(ind[2].mean(0), ref.mean(0), rtol=0.03)
