"""GPU: hand-written adjoints against central finite differences of the forward kernels they differentiate (directional derivatives,
accumulated in float64). The forward kernels themselves are checked against the oracle in test_gpu_passes.py."""
import numpy as np
import pytest

from util import SmallFrame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle, scene_mod):
    import torch
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR
    F = SmallFrame(oracle, scene_mod, fx=40, fy=32)
    W = RR.restirbvhWorker(torch.from_numpy(F.vert).cuda(), torch.from_numpy(F.tri).cuda()); W.update_mesh(W.vrt, W.v_ind)
    mods = RR.load_m_for_restir(F.fx, F.fy)
    return F, W, mods, torch


def _dd(f, x, d, eps):
    """central difference of scalar f along direction d"""
    return (f(x + eps * d) - f(x - eps * d)) / (2 * eps)


def _state(F, oracle, torch):
    O = oracle; N = F.N
    tile_ld, _, tile_pdf = O.light_tiles(F.frame, 300)
    r0 = O.new_reservoirs(N); O.initial(F.frame, r0, tile_ld, tile_pdf, 302)
    vis = O.final_vis(F.frame, r0)
    fdir, fdist, fLi = O.eval_final(F.frame, r0, vis)
    return r0, vis, fdir, fdist, fLi


def test_final_shading_backward(env, oracle):
    F, W, mods, torch = env
    from mirres_restir_nerf_mesh_amd import Resampling as RS
    r0, vis, fdir, fdist, fLi = _state(F, oracle, torch)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    const = dict(fdir=cu(fdir), fdist=cu(fdist[:, None]), tex=cu(F.tex), occ=cu(F.occ[:, None]), rd=cu(F.ray_dir))
    g = torch.Generator(device="cuda").manual_seed(0)
    wts = [torch.rand((F.N, 3), device="cuda", generator=g) for _ in range(3)]

    def fwd(Li, normal, kd, rm):
        c, d, s = RS.FinalShading.apply(mods[6], const["fdir"], const["fdist"], Li, const["tex"], F.Wc, F.Hc, F.fx, F.fy, const["occ"], normal, const["rd"], kd, rm)
        return (c.double() * wts[0]).sum() + (d.double() * wts[1]).sum() + (s.double() * wts[2]).sum()

    x = [cu(fLi).requires_grad_(True), cu(F.normal).requires_grad_(True), cu(F.kd).requires_grad_(True), cu(F.rm).requires_grad_(True)]
    fwd(*x).backward()
    names = ["Li", "normal", "kd", "rough_metal"]
    for i, nm in enumerate(names):
        assert torch.isfinite(x[i].grad).all(), nm
        d = torch.randn(x[i].shape, device="cuda", generator=g)
        if nm == "normal":                       # stay on the unit sphere's tangent plane? not required: the kernel is differentiable off-sphere
            d = d * 0.5
        ana = float((x[i].grad.double() * d).sum())
        args = [t.detach() for t in x]
        def f(v):
            a = list(args); a[i] = v.float().contiguous()
            with torch.no_grad():
                return float(fwd(*a))
        num = _dd(f, args[i].double(), d.double(), 2e-3)
        mag = float((x[i].grad.double() * d).abs().sum())      # the directional derivative is a sum of signed per-element terms: tolerance on their magnitude
        assert abs(ana - num) <= 0.03 * abs(num) + 1e-3 * mag + 1e-2, (nm, ana, num, mag)


def test_eval_final_backward_is_exact_for_linear_env(env, oracle):
    F, W, mods, torch = env
    from mirres_restir_nerf_mesh_amd import Resampling as RS
    r0, vis, _, _, _ = _state(F, oracle, torch)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    res = tuple(cu(a.reshape(F.N, -1)) for a in r0)
    gvis = cu(vis[:, None])
    g = torch.Generator(device="cuda").manual_seed(1)
    w = torch.rand((F.N, 3), device="cuda", generator=g)
    def fwd(tex):
        Li = RS.EvaluateFinalSamples_di.apply(mods[5], res[0], res[1], res[2], res[3], tex, F.Wc, F.Hc, F.fx, F.fy, torch.zeros((F.N, 3), device="cuda"),
                                              torch.zeros((F.N, 1), device="cuda"), gvis)
        return (Li.double() * w).sum()
    tex = cu(F.tex).requires_grad_(True)
    fwd(tex).backward()
    d = torch.rand(tex.shape, device="cuda", generator=g)
    ana = float((tex.grad.double() * d).sum())
    with torch.no_grad():
        num = float(fwd((tex + d).detach()) - fwd(tex.detach()))     # Li is linear in the texels
    assert abs(ana - num) <= 2e-4 * abs(num) + 1e-5, (ana, num)
    assert float(tex.grad.abs().sum()) > 0


def test_eaw_backward(env, oracle):
    F, W, mods, torch = env
    from mirres_restir_nerf_mesh_amd.Denoising import EAWDenoise_run
    g = torch.Generator(device="cuda").manual_seed(2)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    col = torch.rand((F.N, 3), device="cuda", generator=g)
    w = torch.rand((F.N, 3), device="cuda", generator=g)
    occ, nrm, pos = cu(F.occ[:, None]), cu(F.normal), cu(F.pos)
    phis = (2.0, 0.5, 0.05)   # softer n/p kernels than the defaults so that the n/p gradients are not vanishing
    def fwd(c, n, p):
        return (EAWDenoise_run.apply(mods[7], phis[0], phis[1], phis[2], F.fx, F.fy, 2, occ, c, n, p).double() * w).sum()
    x = [col.clone().requires_grad_(True), nrm.clone().requires_grad_(True), pos.clone().requires_grad_(True)]
    fwd(*x).backward()
    for i, (nm, eps) in enumerate((("color", 1e-3), ("normal", 1e-3), ("pos", 2e-4))):
        d = torch.randn(x[i].shape, device="cuda", generator=g)
        ana = float((x[i].grad.double() * d).sum())
        args = [t.detach() for t in x]
        def f(v):
            a = list(args); a[i] = v.float().contiguous()
            with torch.no_grad():
                return float(fwd(*a))
        num = _dd(f, args[i].double(), d.double(), eps)
        assert abs(ana - num) <= 0.03 * abs(num) + 2e-2, (nm, ana, num)
    # the gather form used by EAWDenoise_run.backward and the scatter form (mirres_eaw_bwd) are the same adjoint
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    gout = w.contiguous()
    gs = [torch.zeros_like(t) for t in (col, nrm, pos)]
    check(lib().mirres_eaw_bwd(F.fx, F.fy, 2, phis[0], phis[1], phis[2], occ.data_ptr(), col.data_ptr(), nrm.data_ptr(), pos.data_ptr(), gout.data_ptr(), gs[0].data_ptr(),
                               gs[1].data_ptr(), gs[2].data_ptr(), None), "mirres_eaw_bwd")
    torch.cuda.synchronize()
    for a, b_ in zip(gs, (x[0].grad, x[1].grad, x[2].grad)):
        assert torch.allclose(a, b_, rtol=1e-4, atol=1e-5), float((a - b_).abs().max())


def test_matnet_backward(env, oracle, scene_mod):
    F, W, mods, torch = env
    from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D, GRADIENT_SCALING
    mn, mx = scene_mod.material_min_max(me_max=0.6)
    mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=5)
    with torch.no_grad():
        mlp.encoder.params.mul_(2e3)
    g = torch.Generator(device="cuda").manual_seed(3)
    pts = torch.rand((60000, 3), device="cuda", generator=g) * 1.6 - 0.8      # many points: the fp16 forward makes finite differences noisy
    w = torch.rand((60000, 6), device="cuda", generator=g)
    def loss():
        return (mlp.sample(pts).double() * w).sum()
    loss().backward()
    for i in (0, 2, 4):
        p = mlp.net.net[i].weight
        d = torch.sign(p.grad)                     # steepest direction: largest signal against fp32 round-off of the 360k-term loss
        ana = float((p.grad.double() * d).sum())
        base = p.detach().clone()
        eps = 2e-4
        with torch.no_grad():
            p.copy_(base + eps * d); a = float(loss()); p.copy_(base - eps * d); b = float(loss()); p.copy_(base)
        num = (a - b) / (2 * eps)
        assert abs(ana - num) <= 0.03 * abs(num), (i, ana, num)
    # hash grid: the forward is strongly non-linear (kaiming MLP) and fp16-quantised, so probe along the steepest direction (largest signal)
    # with a small step, on the dense levels
    P = mlp.encoder.params
    d = torch.zeros_like(P); d[:2 * 174880] = torch.sign(P.grad[:2 * 174880])
    ana = float((P.grad.double() * d).sum()) / GRADIENT_SCALING        # the reference's hooks scale the encoder gradient by 128
    base = P.detach().clone()
    eps = 0.0025
    with torch.no_grad():
        P.copy_(base + eps * d); a = float(loss()); P.copy_(base - eps * d); b = float(loss()); P.copy_(base)
    num = (a - b) / (2 * eps)
    assert ana > 0 and abs(ana - num) <= 0.05 * abs(num), (ana, num)


def test_matnet_position_gradient(oracle, scene_mod):
    """MLPTexture3D.sample is differentiable in its argument (tcnn's HashGrid input gradient; the reference routes kd / ks loss gradients to
    `vertices_offsets` through it, nerf/renderer.py:1017-1018).  mirres_matnet_bwd's d/dpos against autograd through a plain-torch fp64 restatement
    of the same field (tests/util.py): inside the AABB, on its faces (torch.clamp passes the gradient at 0 and 1) and outside (blocked)."""
    import torch
    from util import torch_material_field
    from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
    params, w0, w1, w2 = scene_mod.make_matnet_params(seed=4)
    mn, mx = scene_mod.material_min_max(me_max=0.6)
    lo, hi = (-1.0, -2.0, -1.0), (1.0, 2.0, 3.0)                     # anisotropic box: the 1 / (max - min) factor differs per axis
    mlp = MLPTexture3D(torch.tensor(lo + hi, dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=1)
    with torch.no_grad():
        mlp.encoder.params.copy_(torch.from_numpy(params).cuda())
        for i, w in zip((0, 2, 4), (w0, w1, w2)):
            mlp.net.net[i].weight.copy_(torch.from_numpy(w).cuda())
    g = torch.Generator(device="cuda").manual_seed(7)
    n = 20000
    u = torch.rand((n, 3), device="cuda", generator=g) * 1.2 - 0.1              # normalised coordinates in [-0.1, 1.1]: some points outside
    u[:6] = torch.tensor([[0, 0.3, 0.4], [1, 0.5, 0.5], [0.2, 0, 1], [0.5, 0.5, 0.5], [-0.01, 0.5, 0.5], [0.5, 1.01, 0.5]], device="cuda")
    lo_t, hi_t = torch.tensor(lo, device="cuda"), torch.tensor(hi, device="cuda")
    pos = (lo_t + u * (hi_t - lo_t)).contiguous()
    wgt = torch.rand((n, 6), device="cuda", generator=g)
    p1 = pos.clone().requires_grad_(True)
    out = mlp.sample(p1)
    (out * wgt).sum().backward()
    assert p1.grad is not None and p1.grad.shape == (n, 3) and torch.isfinite(p1.grad).all()
    p2 = pos.clone().requires_grad_(True)
    ref = torch_material_field(oracle, params, w0, w1, w2, lo, hi, mn, mx, p2)
    (ref * wgt.double()).sum().backward()
    assert float((out.detach().double() - ref.detach()).abs().max()) < 2e-4           # forward: fp16 interpolation against fp64
    got, want = p1.grad.double(), p2.grad
    xn = (pos.double() - lo_t.double()) / (hi_t.double() - lo_t.double())
    outside = (xn < 0) | (xn > 1)
    assert outside.any() and bool((got[outside] == 0).all()) and bool((want[outside] == 0).all())
    assert bool((got[1, 0] != 0)) and bool((got[0, 0] != 0))                             # on the faces x = 1 / x = 0 the gradient passes (clamp is inclusive)
    scale = want.abs().mean()
    err = (got - want).abs()
    # the forward's fp16 interpolation (relative 1e-3 on a feature) flips a ReLU that sits near zero — 64 hidden units per point, so about one
    # point in a hundred differs by a whole unit's contribution; everything else agrees to the rounding of the fp32 sums
    ok = err <= 2e-3 * want.abs() + 2e-3 * scale
    assert float(ok.double().mean()) > 0.985, (float(ok.double().mean()), float(err.max()), float(scale))
    assert float((got * want).sum() / (want * want).sum()) == pytest.approx(1.0, abs=1e-2)
    # a detached argument gets no gradient buffer and the parameter gradients are unchanged by asking for d/dpos
    mlp.zero_grad(); (mlp.sample(pos) * wgt).sum().backward(); gp0 = mlp.encoder.params.grad.clone(); gw0 = mlp.net.net[0].weight.grad.clone()
    mlp.zero_grad(); p3 = pos.clone().requires_grad_(True); (mlp.sample(p3) * wgt).sum().backward()
    assert torch.allclose(mlp.encoder.params.grad, gp0, rtol=1e-4, atol=1e-6 * float(gp0.abs().max())) and torch.allclose(mlp.net.net[0].weight.grad, gw0, rtol=1e-3, atol=1e-5 * float(gw0.abs().max()))
