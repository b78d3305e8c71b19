"""GPU parity (through the C ABI / the reference-shaped Python surface): each ReSTIR / shading pass against the oracle on identical inputs.

Everything is compared BIT FOR BIT: integer state (RNG streams, selections, M) and every float.  Product and oracle share the FP policy (fp32, IEEE
division / square root, no contraction) and, since round 3, the arithmetic of the transcendental functions (include/mirres_fmath.h; device and host
bits compared exhaustively in test_gpu_fmath.py), so no discrete choice (CDF bin, reservoir selection, lobe) can differ and no float either."""
import numpy as np
import pytest

from util import SmallFrame, match_fraction, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle, scene_mod):
    import torch
    assert torch.cuda.is_available()
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR
    F = SmallFrame(oracle, scene_mod)
    W = RR.restirbvhWorker(torch.from_numpy(F.vert).cuda(), torch.from_numpy(F.tri).cuda())
    W.update_mesh(W.vrt, W.v_ind)
    mods = RR.load_m_for_restir(F.fx, F.fy)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    T = dict(occ=cu(F.occ[:, None]), pos=cu(F.pos), nd=cu(F.normal_depth), brdf=cu(F.brdf), rd=cu(F.ray_dir), tex=cu(F.tex), normal=cu(F.normal), kd=cu(F.kd),
             rm=cu(F.rm), pdf=cu(F.tables[0]), cdf=cu(F.tables[1]), mpdf=cu(F.tables[2]), mcdf=cu(F.tables[3]), noff=cu(F.noff))
    return F, W, mods, T, torch


def _res_np(res):
    return [r.detach().cpu().numpy().reshape(r.shape[0], -1).squeeze() for r in res]


def _new_res(torch, N):
    return (torch.zeros((N, 3), device="cuda"), torch.zeros((N, 1), device="cuda"), torch.zeros((N, 1), dtype=torch.int32, device="cuda"), torch.zeros((N, 1), device="cuda"))


def _cmp_res(gpu, ref, what):
    g = _res_np(gpu)
    assert np.array_equal(g[2], ref[2]), what + ": M"
    for k, nm in ((0, "light_data"), (1, "light_pdf"), (3, "weight")):
        same_bits(g[k], ref[k], "%s reservoirs / %s" % (what, nm))
    return np.ones(len(g[2]), bool)


def test_tables_and_tiles(env, oracle):
    F, W, mods, T, torch = env
    from mirres_restir_nerf_mesh_amd.GenerateLightTiles import make_sampleable, GenerateLightTiles
    pdf, cdf, mpdf, mcdf = make_sampleable(mods[0], T["tex"], F.Wc, F.Hc)
    for g, r, nm in zip((pdf, cdf, mpdf, mcdf), F.tables, ("pdf", "cdf", "mpdf", "mcdf")):
        same_bits(g.cpu().numpy().ravel(), r, "importance tables / " + nm)
    assert float(cdf.view(F.Hc, F.Wc + 1)[:, -1].min()) == 1.0 and float(mcdf[-1]) == 1.0
    ld, uv, ip = mods[8], mods[9], mods[10]
    GenerateLightTiles(mods[1], None, T["tex"], T["pdf"], T["cdf"], T["mpdf"], T["mcdf"], F.Wc, F.Hc, 777, ld, uv, ip)
    rld, ruv, rip = oracle.light_tiles(F.frame, 777)
    same_bits(ld.cpu().numpy(), rld, "light tiles / light_data")
    assert np.array_equal(uv.cpu().numpy(), ruv)
    same_bits(ip.cpu().numpy().ravel(), rip, "light tiles / pdf")
    # tiles 64..127 duplicate tiles 0..63: 16-bit seed masking + scalar splat — size-independent property
    l = ld.cpu().numpy().reshape(128, 1024, 3)
    assert np.array_equal(l[:64], l[64:])
    no = mods[14].cpu().numpy()
    assert np.array_equal(no, F.noff)
    assert np.array_equal((no[:6] * 127).round().astype(int), np.array([[-62, -109], [67, -73], [4, 70], [-57, -38], [72, -2], [9, -112]]))


def test_reservoir_chain(env, oracle):
    """initial -> temporal -> spatial -> vis -> eval_final -> final_shading, each fed with the ORACLE's previous state."""
    F, W, mods, T, torch = env
    from mirres_restir_nerf_mesh_amd import Resampling as RS
    O = oracle; N = F.N
    tile_ld, _, tile_pdf = O.light_tiles(F.frame, 1000)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    g_tile_ld, g_tile_pdf = cu(tile_ld), cu(tile_pdf)
    # --- initial (frame 0) and a second initial (frame 1) to have a history
    r0 = O.new_reservoirs(N); O.initial(F.frame, r0, tile_ld, tile_pdf, 1002)
    g0 = _new_res(torch, N)
    W.InitialResampling_(mods[2], T["pos"], g0, T["tex"], F.Wc, F.Hc, F.fx, F.fy, 1002, T["occ"], T["nd"], T["brdf"], T["rd"], T["pdf"], T["cdf"], T["mpdf"], T["mcdf"],
                         g_tile_ld, None, g_tile_pdf)
    ok = _cmp_res(g0, r0, "initial")
    fg = F.occ > 0.5
    assert (r0[2][fg] == 1).all() and (r0[2][~fg] == 0).all()      # M := 1 on surfaces, 0 on background
    assert (_res_np(g0)[2] == r0[2]).all()
    # --- temporal: current = r1, previous = r0 (oracle states on both sides)
    r1 = O.new_reservoirs(N); O.initial(F.frame, r1, tile_ld, tile_pdf, 1022)
    ref = [a.copy() for a in r1]
    O.temporal(F.frame, ref, r0, F.occ, F.normal_depth, F.brdf, F.ray_dir, 1023)
    g1 = tuple(cu(a.reshape(N, -1)) for a in r1); gp = tuple(cu(a.reshape(N, -1)) for a in r0)
    RS.TemporalResampling(mods[3], g1, gp, T["tex"], F.Wc, F.Hc, F.fx, F.fy, 1023, T["occ"], T["nd"], T["brdf"], T["rd"], T["occ"], T["nd"], T["brdf"], T["rd"], None)
    _cmp_res(g1, ref, "temporal")
    assert ref[2].max() == 2 and (ref[2][fg] >= 1).all()
    # --- spatial: prev = ref (post-temporal), out = new
    sref = O.new_reservoirs(N); cnt = np.zeros(4, np.uint64)
    O.spatial(F.frame, sref, ref, F.noff, 1024, cnt)
    gs = _new_res(torch, N); gprev = tuple(cu(a.reshape(N, -1)) for a in ref)
    W.SpatialResampling_(mods[4], T["pos"], gs, gprev, T["noff"], T["tex"], F.Wc, F.Hc, F.fx, F.fy, 1024, T["occ"], T["nd"], T["brdf"], T["rd"])
    _cmp_res(gs, sref, "spatial")
    # --- visibility of the final sample: bit-exact given identical reservoirs
    vis_ref = O.final_vis(F.frame, sref)
    gvis = torch.ones((N, 1), device="cuda"); gsr = tuple(cu(a.reshape(N, -1)) for a in sref)
    W.EvaluateFinalSamples_get_vis(mods[5], T["pos"], gsr, F.fx, F.fy, gvis)
    assert np.array_equal(gvis.cpu().numpy().ravel(), vis_ref)
    # --- eval_final + final shading
    fdir, fdist, fLi = O.eval_final(F.frame, sref, vis_ref)
    gdir = torch.zeros((N, 3), device="cuda"); gdist = torch.zeros((N, 1), device="cuda")
    gLi = RS.EvaluateFinalSamples_di.apply(mods[5], gsr[0], gsr[1], gsr[2], gsr[3], T["tex"], F.Wc, F.Hc, F.fx, F.fy, gdir, gdist, cu(vis_ref[:, None]))
    same_bits(gdir.cpu().numpy(), fdir, "final sample / direction")
    assert np.array_equal(gdist.cpu().numpy().ravel(), fdist)
    same_bits(gLi.cpu().numpy(), fLi, "final sample / Li")
    c, d, s = O.final_shading(F.frame, F.normal, F.kd, F.rm, fdir, fdist, fLi)
    gc, gd, gsx = RS.FinalShading.apply(mods[6], cu(fdir), cu(fdist[:, None]), cu(fLi), T["tex"], F.Wc, F.Hc, F.fx, F.fy, T["occ"], T["normal"], T["rd"], T["kd"], T["rm"])
    for a, b, nm in ((gc, c, "color"), (gd, d, "diff"), (gsx, s, "spec")):
        same_bits(a.cpu().numpy(), b, "FinalShading / " + nm)   # north-star bar: 1e-3 per channel abs — met with zero error


def test_reservoir_passes_with_other_constants(env, oracle, scene_mod):
    """The ReSTIR constants are runtime configuration on both sides (mirres_config_t / the oracle's Config): 16 light candidates, a history cap of 7
    and SEVEN spatial neighbours — more than five selects the second instantiation of the spatial kernels (k_spatial_gen<8> / k_spatial_resolve<8>),
    which no other test reaches. initial -> temporal -> spatial against the oracle, as test_reservoir_chain does for the defaults."""
    F, W, _, T, torch = env
    from mirres_restir_nerf_mesh_amd import Resampling as RS, _lib, _ops
    O = oracle; N = F.N
    fr = O.make_frame(F.keep, F.fx, F.fy, F.occ, F.pos, F.normal_depth, F.brdf, F.ray_dir, (F.info, F.aabb), F.vert, F.tri, F.tex, F.Wc, F.Hc, F.tables,
                      neighbor_count=7, initial_light_samples=16, max_history=7)
    cfg = _lib.default_config(); cfg.neighbor_count, cfg.initial_light_samples, cfg.max_history = 7, 16, 7
    m = _ops.Module("restir (non-default constants)", _ops.Context(F.fx, F.fy, cfg))
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tile_ld, _, tile_pdf = O.light_tiles(fr, 2000)
    g_tile_ld, g_tile_pdf = cu(tile_ld), cu(tile_pdf)
    r0 = O.new_reservoirs(N); O.initial(fr, r0, tile_ld, tile_pdf, 2002)
    g0 = _new_res(torch, N)
    W.InitialResampling_(m, T["pos"], g0, T["tex"], F.Wc, F.Hc, F.fx, F.fy, 2002, T["occ"], T["nd"], T["brdf"], T["rd"], T["pdf"], T["cdf"], T["mpdf"], T["mcdf"],
                         g_tile_ld, None, g_tile_pdf)
    _cmp_res(g0, r0, "initial, 16 candidates")
    ref_default = O.new_reservoirs(N); O.initial(F.frame, ref_default, tile_ld, tile_pdf, 2002)
    assert not np.array_equal(ref_default[0], r0[0])          # the constant reaches the oracle: 32 candidates select differently
    # temporal: several rounds so that the history cap binds (M <= 7 x current M)
    prev = [a.copy() for a in r0]
    for it in range(9):
        cur = O.new_reservoirs(N); O.initial(fr, cur, tile_ld, tile_pdf, 2100 + 2 * it)
        ref = [a.copy() for a in cur]
        O.temporal(fr, ref, prev, F.occ, F.normal_depth, F.brdf, F.ray_dir, 2101 + 2 * it)
        if it == 8:
            g1 = tuple(cu(a.reshape(N, -1)) for a in cur); gp = tuple(cu(a.reshape(N, -1)) for a in prev)
            RS.TemporalResampling(m, g1, gp, T["tex"], F.Wc, F.Hc, F.fx, F.fy, 2101 + 2 * it, T["occ"], T["nd"], T["brdf"], T["rd"], T["occ"], T["nd"], T["brdf"], T["rd"], None)
            _cmp_res(g1, ref, "temporal, history cap 7")
        prev = ref
    assert prev[2].max() == 8 and prev[2].max() < 9           # 1 + min(M_prev, 7): the cap of 7 binds (the default of 20 would allow 10 here)
    # spatial with seven neighbours
    sref = O.new_reservoirs(N); cnt = np.zeros(4, np.uint64)
    O.spatial(fr, sref, prev, F.noff, 2200, cnt)
    s5 = O.new_reservoirs(N); O.spatial(F.frame, s5, prev, F.noff, 2200, np.zeros(4, np.uint64))
    assert not np.array_equal(s5[3], sref[3])                 # seven neighbours merge differently from five
    gs = _new_res(torch, N); gprev = tuple(cu(a.reshape(N, -1)) for a in prev)
    W.SpatialResampling_(m, T["pos"], gs, gprev, T["noff"], T["tex"], F.Wc, F.Hc, F.fx, F.fy, 2200, T["occ"], T["nd"], T["brdf"], T["rd"])
    _cmp_res(gs, sref, "spatial, 7 neighbours")


def test_path_vertices(env, oracle):
    """new_dir + two bounce kernels on oracle state; hit / stop flags exact, radiance within tolerance."""
    F, W, mods, T, torch = env
    from mirres_restir_nerf_mesh_amd import Resampling as RS
    O = oracle; N = F.N
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    z = lambda *s: np.zeros(s, np.float32)
    prd, npos, nrd, nocc, nn = z(N, 5), z(N, 3), z(N, 3), z(N), z(N, 3)
    P = O.path_struct(F.occ, F.pos, F.normal, F.ray_dir, F.kd, F.rm, prd, npos, nrd, nocc, nn)
    O.new_dir(F.frame, P, 2004, 0)
    gprd, gnpos, gnrd, gnocc, gnn = (torch.zeros(s, device="cuda") for s in ((N, 5), (N, 3), (N, 3), (N, 1), (N, 3)))
    RS.process_new_dir_for_pt(mods[6], W.LBVHNode_info, W.LBVHNode_aabb, W.vrt, W.v_ind, 2004, 0, F.fx, F.fy, T["occ"], T["pos"], T["normal"], T["rd"], gprd,
                              T["kd"], T["rm"], gnpos, gnrd, gnocc, gnn)
    same_bits(gprd.cpu().numpy(), prd, "new_dir / prd")
    assert np.array_equal(gnocc.cpu().numpy().ravel(), nocc)
    hitm = nocc > 0.5
    same_bits(gnpos.cpu().numpy()[hitm], npos[hitm], "new_dir / hit point")
    same_bits(gnrd.cpu().numpy(), nrd, "new_dir / direction")
    same_bits(gnn.cpu().numpy()[hitm], nn[hitm], "new_dir / hit normal")
    assert hitm.sum() > 100
    # bounce 1 on the oracle's vertex state, constant material at the hits
    kd1 = np.where(nocc[:, None] >= 0.5, np.float32(0.55), np.float32(0)).astype(np.float32) * np.ones((1, 3), np.float32)
    rm1 = np.zeros((N, 2), np.float32); rm1[nocc >= 0.5] = (0.4, 0.1)
    prd_in = prd.copy()
    tpos, trd, tocc, tn = z(N, 3), z(N, 3), z(N), z(N, 3)
    P1 = O.path_struct(nocc, npos, nn, nrd, kd1, rm1, prd, tpos, trd, tocc, tn)
    c, d, s = O.bounce(F.frame, P1, 2009, 1)
    gprd = cu(prd_in); gt = [torch.zeros(sh, device="cuda") for sh in ((N, 3), (N, 3), (N, 1), (N, 3))]
    gc, gd, gs = (torch.zeros((N, 3), device="cuda") for _ in range(3))
    RS.indirect_one_hit_divided_no_grad(mods[6], W.LBVHNode_info, W.LBVHNode_aabb, W.vrt, W.v_ind, 2009, 1, F.fx, F.fy, T["tex"], F.Wc, F.Hc, T["pdf"], T["cdf"],
                                        T["mpdf"], T["mcdf"], cu(nocc[:, None]), cu(npos), cu(nn), cu(nrd), gprd, cu(kd1), cu(rm1), gc, gd, gs, gt[0], gt[1], gt[2], gt[3])
    for a, b, nm in ((gc, c, "color"), (gd, d, "diffuse"), (gs, s, "specular")):
        same_bits(a.cpu().numpy(), b, "indirect vertex / " + nm)
    same_bits(gprd.cpu().numpy(), prd, "indirect vertex / prd")
    assert np.array_equal(gt[2].cpu().numpy().ravel(), tocc)
    assert c[nocc > 0.5].mean() > 0 and tocc.sum() > 50


def test_eaw(env, oracle):
    F, W, mods, T, torch = env
    from mirres_restir_nerf_mesh_amd.Denoising import EAWDenoise_run_no_di
    rng = np.random.default_rng(5)
    col = (rng.random((F.N, 3)) * 2).astype(np.float32)
    for step in (2, 1):
        ref = oracle.eaw(F.fx, F.fy, step, 2.0, 0.1, 0.001, F.occ, col, F.normal, F.pos)
        out = EAWDenoise_run_no_di(mods[7], 2.0, 0.1, 0.001, F.fx, F.fy, step, T["occ"], torch.from_numpy(col).cuda(), T["normal"], T["pos"])
        same_bits(out.cpu().numpy(), ref, "a-trous step %d" % step)
    # constant image is a fixed point on foreground pixels (size-independent property)
    one = torch.full((F.N, 3), 0.37, device="cuda")
    out = EAWDenoise_run_no_di(mods[7], 2.0, 0.1, 0.001, F.fx, F.fy, 2, T["occ"], one, T["normal"], T["pos"])
    np.testing.assert_allclose(out.cpu().numpy(), 0.37, rtol=1e-6)


def test_eaw_driver_matches_the_reference_driver(env):
    """EAWDenoise_use_phi / _no_di of the package (HIP kernel through the C ABI) against the fixture produced by the REFERENCE's own Denoising.py
    driver run over the oracle kernel (tests/golden/ref_python.npz): pins the step schedule 2, 1 / 4, 2, 1 and the buffer hand-over."""
    import os
    F, W, mods, T, torch = env
    from mirres_restir_nerf_mesh_amd.Denoising import EAWDenoise_use_phi, EAWDenoise_use_phi_no_di
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python.npz"))
    fx, fy = [int(v) for v in g["eaw_dims"]]
    t = lambda k: torch.from_numpy(g[k]).cuda()
    a = EAWDenoise_use_phi(mods[7], 2.0, 0.1, 0.001, 2, 2, fx, fy, t("eaw_occ"), t("eaw_col"), t("eaw_nrm"), t("eaw_pos"))
    b = EAWDenoise_use_phi_no_di(mods[7], 2.0, 0.1, 0.001, 4, 3, fx, fy, t("eaw_occ"), t("eaw_col"), t("eaw_nrm"), t("eaw_pos"))
    np.testing.assert_allclose(a.cpu().numpy(), g["eaw_di"], rtol=3e-5, atol=2e-6)
    np.testing.assert_allclose(b.cpu().numpy(), g["eaw_nodi"], rtol=3e-5, atol=2e-6)


def test_render_finish_matches_the_reference_post_processing(env):
    """mirres_render_finish (average, five a-trous runs, composite, background, nan_to_num) on prepared sums against the fixture produced by the
    REFERENCE's own run_restir_di_with_pt (executed from its AST with the spp loop replaced by those sums; tests/golden/ref_python.npz)."""
    import os, ctypes as C
    F, W, mods, T, torch = env
    from mirres_restir_nerf_mesh_amd import dist as D
    from mirres_restir_nerf_mesh_amd._ops import get_ctx
    from mirres_restir_nerf_mesh_amd._lib import lib, check, stream_ptr
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python.npz"))
    fx, fy = [int(v) for v in g["eaw_dims"]]; N = fx * fy
    ctx = get_ctx(fx, fy)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    gb = dict(occ=c(g["fin_occ"]), normal=c(g["eaw_nrm"]), depth=torch.zeros((N, 1), device="cuda"), kd=c(g["fin_kd"]), rm=c(g["fin_rm"]), ray_dir=c(g["fin_rd"]), pos=c(g["eaw_pos"]))
    _, a, keep = D._finish_args(ctx, torch.full((8, 16, 3), 0.5, device="cuda"), gb, int(g["fin_spp"]), 2, 2, 2.0, 0.1, 0.001)
    sums = [c(g["fin_sums"][k]) for k in range(6)]
    outs = [torch.empty((N, 3), device="cuda") for _ in range(6)]
    for k in range(6):
        a.outs[k] = outs[k].data_ptr()
    arr = (C.c_void_p * 6)(*[s_.data_ptr() for s_ in sums])
    check(lib().mirres_render_finish(ctx.h, C.byref(a), arr, stream_ptr()), "mirres_render_finish")
    for k in range(6):
        np.testing.assert_allclose(outs[k].cpu().numpy(), g["fin_out"][k], rtol=3e-5, atol=3e-6, err_msg=str(k))


def test_normal_ao_matches_the_oracle_and_the_reference_call_shape(env, oracle):
    """process_normal_ao on the small frame's G-buffer, launched the way nerf/renderer.py:1153-1158 launches it off the denoising module handle
    (`m.process_normal_ao(framedim_x=..., ..., out_ao=...).launchRaw(blockSize=..., gridSize=...)`): bit-equal to the oracle (sums of clamped dot
    products in the same order), zero on the background; wrongly sized tensors are refused."""
    F, W, mods, T, torch = env
    from mirres_restir_nerf_mesh_amd._lib import MirresError
    denoising_m = mods[7]
    occ = T["occ"]; nrm = T["normal"]; rd = T["rd"]
    out_ao = torch.full((F.N, 3), -1.0, device="cuda")
    denoising_m.process_normal_ao(framedim_x=int(F.fx), framedim_y=int(F.fy), occ_map=occ, normal_map=nrm.detach(), ray_dir=rd, out_ao=out_ao) \
        .launchRaw(blockSize=(16, 16, 1), gridSize=((int(F.fx) + 15) // 16, (int(F.fy) + 15) // 16, 1))
    ref = oracle.normal_ao(F.fx, F.fy, F.occ, F.normal)
    got = out_ao.cpu().numpy()
    assert np.array_equal(got, ref)
    assert np.all(got[F.occ < 0.1] == 0) and got.max() > 0 and got.min() >= 0 and got.max() <= 1
    with pytest.raises(MirresError):
        denoising_m.process_normal_ao(framedim_x=int(F.fx), framedim_y=int(F.fy), occ_map=occ, normal_map=nrm[:-1], ray_dir=rd, out_ao=out_ao)
