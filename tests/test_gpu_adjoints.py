"""GPU: the hand-written adjoints (csrc/backward.hip, eaw.hip) against AUTOGRAD of the oracle's formulas, ELEMENT BY ELEMENT — parity, not a property test.

The reference obtains these derivatives from Slang's automatic differentiation of process_FinalShading / process_EvaluateFinalSamples_di_ / process_EAWDenoise
(Resampling.py:119-214, Denoising.py:30-48) and from torch / tcnn autograd of the material field.  The checker here is tests/adjoint_refs.py: the same formulas as
the CPU oracle, restated in float64 torch so that autograd differentiates them; each test first holds that restatement's forward to the ORACLE's output on the
same inputs (<= 2e-5: it is the oracle's function), then compares every gradient element of every input with the HIP adjoint (rtol 1e-3 plus an absolute term of
1e-4 of the gradient's scale: fp32 adjoint arithmetic against fp64)."""
import numpy as np
import pytest

from util import SmallFrame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle, scene_mod):
    import torch
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR
    F = SmallFrame(oracle, scene_mod, fx=72, fy=60)        # ~1 900 foreground pixels
    W = RR.restirbvhWorker(torch.from_numpy(F.vert).cuda(), torch.from_numpy(F.tri).cuda()); W.update_mesh(W.vrt, W.v_ind)
    mods = RR.load_m_for_restir(F.fx, F.fy)
    O = oracle
    tile_ld, _, tile_pdf = O.light_tiles(F.frame, 300)
    r0 = O.new_reservoirs(F.N); O.initial(F.frame, r0, tile_ld, tile_pdf, 302)
    vis = O.final_vis(F.frame, r0)
    fdir, fdist, fLi = O.eval_final(F.frame, r0, vis)
    return F, W, mods, torch, dict(res=r0, vis=vis, fdir=fdir, fdist=fdist, fLi=fLi)


def _elementwise(got, want, what, rtol=1e-3, atol_scale=1e-4, min_frac=1.0):
    got = got.detach().double().cpu().numpy(); want = want.detach().double().cpu().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(np.abs(want).max())
    assert scale > 0, what + ": reference gradient is identically zero"
    err = np.abs(got - want)
    ok = err <= rtol * np.abs(want) + atol_scale * scale
    assert ok.mean() >= min_frac, "%s: %d of %d gradient elements differ (max err %.3e at reference %.3e, scale %.3e)" % (
        what, int((~ok).sum()), ok.size, float(err.max()), float(np.abs(want).ravel()[np.argmax(err)]), scale)


def test_final_shading_adjoint_element_by_element(env, oracle):
    F, W, mods, torch, st = env
    import adjoint_refs as R
    from mirres_restir_nerf_mesh_amd import Resampling as RS
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    occ, rd, fdir, fdist = cu(F.occ[:, None]), cu(F.ray_dir), cu(st["fdir"]), cu(st["fdist"][:, None])
    g = torch.Generator(device="cuda").manual_seed(0)
    wts = [torch.rand((F.N, 3), device="cuda", generator=g) for _ in range(3)]
    x32 = [cu(a).requires_grad_(True) for a in (F.normal, F.kd, F.rm, st["fLi"])]
    c, d, s = RS.FinalShading.apply(mods[6], fdir, fdist, x32[3], cu(F.tex), F.Wc, F.Hc, F.fx, F.fy, occ, x32[0], rd, x32[1], x32[2])
    ((c * wts[0]).sum() + (d * wts[1]).sum() + (s * wts[2]).sum()).backward()
    x64 = [cu(a).double().requires_grad_(True) for a in (F.normal, F.kd, F.rm, st["fLi"])]
    c64, d64, s64 = R.final_shading(occ.double(), x64[0], rd.double(), x64[1], x64[2], fdir.double(), fdist.double(), x64[3])
    # the restatement IS the oracle's function: its forward equals the oracle's (and the HIP kernel's, bit-equal to the oracle elsewhere) on foreground pixels
    oc, od, os_ = oracle.final_shading(F.frame, F.normal, F.kd, F.rm, st["fdir"], st["fdist"], st["fLi"])
    fg = F.occ > 0.5
    for a, b in ((c64, oc), (d64, od), (s64, os_)):
        np.testing.assert_allclose(a.detach().cpu().numpy()[fg], b[fg], rtol=5e-4, atol=2e-6)       # fp64 against fp32: (c a^2 - c) c + 1 of the GGX lobe cancels for small alpha near the highlight
    ((c64 * wts[0].double()).sum() + (d64 * wts[1].double()).sum() + (s64 * wts[2].double()).sum()).backward()
    assert int(fg.sum()) >= 1000
    fgt = torch.from_numpy(fg).cuda()
    for nm, a, b in zip(("normal", "kd", "rough_metal", "Li"), x32, x64):
        assert torch.isfinite(a.grad).all(), nm
        _elementwise(a.grad[fgt], b.grad[fgt], "FinalShading d/d" + nm)
        assert float(a.grad[~fgt].abs().sum()) == 0.0, nm + ": background pixels carry no gradient"


def test_eval_final_adjoint_element_by_element(env, oracle):
    F, W, mods, torch, st = env
    import adjoint_refs as R
    from mirres_restir_nerf_mesh_amd import Resampling as RS
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    res = tuple(cu(a.reshape(F.N, -1)) for a in st["res"])
    gvis = cu(st["vis"][:, None])
    g = torch.Generator(device="cuda").manual_seed(1)
    w = torch.rand((F.N, 3), device="cuda", generator=g)
    tex = cu(F.tex).requires_grad_(True)
    Li = RS.EvaluateFinalSamples_di.apply(mods[5], res[0], res[1], res[2], res[3], tex, F.Wc, F.Hc, F.fx, F.fy, torch.zeros((F.N, 3), device="cuda"),
                                          torch.zeros((F.N, 1), device="cuda"), gvis)
    (Li * w).sum().backward()
    tex64 = cu(F.tex).double().requires_grad_(True)
    Li64 = R.eval_final(tex64, F.Wc, F.Hc, res[0].double(), res[3].double(), gvis.double())
    np.testing.assert_allclose(Li64.detach().cpu().numpy(), st["fLi"], rtol=3e-5, atol=3e-6)          # = the oracle's Li
    (Li64 * w.double()).sum().backward()
    assert int((tex64.grad.abs().sum(1) > 0).sum()) > 20                                              # a few dozen texels receive the samples (sun lobe)
    _elementwise(tex.grad, tex64.grad, "EvaluateFinalSamples_di d/d(env texel)")


def test_eaw_adjoint_element_by_element(env, oracle):
    F, W, mods, torch, st = env
    import adjoint_refs as R
    from mirres_restir_nerf_mesh_amd.Denoising import EAWDenoise_run
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    g = torch.Generator(device="cuda").manual_seed(2)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    col = torch.rand((F.N, 3), device="cuda", generator=g)
    w = torch.rand((F.N, 3), device="cuda", generator=g)
    occ, nrm, pos = cu(F.occ[:, None]), cu(F.normal), cu(F.pos)
    for step, phis in ((2, (2.0, 0.5, 0.05)), (1, (2.0, 0.1, 0.001))):      # softer kernels (all three gradients alive) and the frame's defaults
        x = [col.clone().requires_grad_(True), nrm.clone().requires_grad_(True), pos.clone().requires_grad_(True)]
        out = EAWDenoise_run.apply(mods[7], phis[0], phis[1], phis[2], F.fx, F.fy, step, occ, *x)
        (out * w).sum().backward()
        x64 = [t.detach().double().requires_grad_(True) for t in x]
        out64 = R.eaw(F.fx, F.fy, step, *phis, occ.double(), *x64)
        ref = oracle.eaw(F.fx, F.fy, step, *phis, F.occ, col.cpu().numpy(), F.normal, F.pos)
        np.testing.assert_allclose(out64.detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-6)           # = the oracle's filter
        (out64 * w.double()).sum().backward()
        for nm, a, b in zip(("colour", "normal", "position"), x, x64):
            if float(b.grad.abs().max()) < 1e-12:
                assert float(a.grad.abs().max()) < 1e-9, nm
                continue
            _elementwise(a.grad, b.grad, "EAW step %d d/d%s" % (step, nm), rtol=2e-3, atol_scale=2e-4)
        # the scatter form of the same adjoint (mirres_eaw_bwd), element by element as well
        gs = [torch.zeros_like(t) for t in (col, nrm, pos)]
        check(lib().mirres_eaw_bwd(F.fx, F.fy, step, phis[0], phis[1], phis[2], occ.data_ptr(), col.data_ptr(), nrm.data_ptr(), pos.data_ptr(), w.contiguous().data_ptr(),
                                   gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr(), None), "mirres_eaw_bwd")
        torch.cuda.synchronize()
        _elementwise(gs[0], x64[0].grad, "EAW step %d scatter d/dcolour" % step, rtol=2e-3, atol_scale=2e-4)


def test_material_field_adjoint_element_by_element(oracle, scene_mod):
    """mirres_matnet_bwd (MLP weights, hash-grid table) against float64 autograd of the field (tests/util.py:torch_material_field, held to the oracle's forward).
    Weight gradients are sums over all points: every element.  The table gradient is sparse: every touched entry."""
    import torch
    from util import torch_material_field
    from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D, GRADIENT_SCALING
    params, w0, w1, w2 = scene_mod.make_matnet_params(seed=4)
    mn, mx = scene_mod.material_min_max(me_max=0.6)
    lo, hi = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    mlp = MLPTexture3D(torch.tensor(lo + hi, dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=1)
    with torch.no_grad():
        mlp.encoder.params.copy_(torch.from_numpy(params).cuda())
        for i, w in zip((0, 2, 4), (w0, w1, w2)):
            mlp.net.net[i].weight.copy_(torch.from_numpy(w).cuda())
    g = torch.Generator(device="cuda").manual_seed(11)
    n = 4000
    pos = (torch.rand((n, 3), device="cuda", generator=g) * 1.6 - 0.8).contiguous()
    wgt = torch.rand((n, 6), device="cuda", generator=g)
    out = mlp.sample(pos)
    (out * wgt).sum().backward()
    keep = oracle.Keep()
    om = oracle.matnet_struct(keep, params, w0, w1, w2, lo, hi, mn, mx)
    assert np.array_equal(out.detach().cpu().numpy(), oracle.matnet(om, pos.cpu().numpy()))                 # the forward is the oracle's, bit for bit
    P64 = torch.from_numpy(oracle.to_f16_bits(params).view(np.float16).astype(np.float64)).cuda().requires_grad_(True)
    W64 = [torch.from_numpy(a.astype(np.float64)).cuda().requires_grad_(True) for a in (w0, w1, w2)]
    ref = torch_material_field(oracle, params, W64[0], W64[1], W64[2], lo, hi, mn, mx, pos.double(), table=P64)
    assert float((ref.detach() - out.detach().double()).abs().max()) < 2e-4                                     # fp64 interpolation against fp16
    (ref * wgt.double()).sum().backward()
    # MLP weights: a ReLU that sits within the fp16 interpolation error of zero is on in one evaluation and off in the other for about one point in a hundred;
    # over 4 000 points that moves a weight's gradient by a fraction of a per cent
    for i, a in zip((0, 2, 4), W64):
        _elementwise(mlp.net.net[i].weight.grad, a.grad, "material MLP d/dW%d" % (i // 2), rtol=2e-2, atol_scale=5e-3)
    got = mlp.encoder.params.grad.double() / GRADIENT_SCALING                                                   # the reference's hook scales the encoder gradient by 128
    want = P64.grad.reshape(-1)
    touched = want != 0
    assert int(touched.sum()) > 10000 and float(got[~touched].abs().max()) == 0.0
    _elementwise(got[touched], want[touched], "hash-grid table gradient", rtol=5e-2, atol_scale=5e-3, min_frac=0.99)
