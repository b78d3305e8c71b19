"""Stage-1 loss / optimiser glue against the reference's own functions (tests/golden/ref_losses.npz, produced by
tests/golden/gen_reference_losses.py from nerf/utils.py), plus closed-form cases for the two pytorch3d-defined regularisers."""
import os
import types

import numpy as np
import pytest
import torch

from mirres_restir_nerf_mesh_amd import losses

G = os.path.join(os.path.dirname(__file__), "golden", "ref_losses.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(G)


def test_shading_loss_value_and_gradients(g):
    """nerf/utils.py:306-318 on 4096 pixels spanning both branches of both tone curves, the log clip and the eps clamps: value to 1e-6 relative,
    input gradients to 1e-6 of their scale."""
    d = torch.from_numpy(g["sh_d"]).requires_grad_(True); s = torch.from_numpy(g["sh_s"]).requires_grad_(True)
    l = losses.shading_loss(d, s, torch.from_numpy(g["sh_ref"]), 0.0015, 0.000025)
    l.backward()
    assert abs(float(l.detach()) - float(g["sh_loss"])) <= 1e-6 * abs(float(g["sh_loss"]))
    for mine, ref in ((d.grad.numpy(), g["sh_gd"]), (s.grad.numpy(), g["sh_gs"])):
        np.testing.assert_allclose(mine, ref, rtol=2e-5, atol=1e-6 * float(np.abs(ref).max()))


def test_material_smoothness_terms(g):
    kd = torch.from_numpy(g["ms_kd"])
    l = losses.material_smoothness_grad(kd, torch.from_numpy(g["ms_ks"]), torch.from_numpy(g["ms_nrm"]), lambda_kd=0.005, lambda_ks=0.0025, lambda_nrm=0.00025)
    assert abs(float(l.detach()) - float(g["ms_loss"])) <= 1e-6 * abs(float(g["ms_loss"]))
    e = losses.material_extra_kd_smoothness_grad(kd, torch.from_numpy(g["ms_ao"]), 0.3)
    assert abs(float(e.detach()) - float(g["ms_extra"])) <= 1e-6 * abs(float(g["ms_extra"]))


def test_uniform_laplacian_value_and_gradient(g):
    """laplacian_smooth_loss (nerf/utils.py:231-274, sparse D - A) on the synthetic mesh (closed part + open floor) with seeded offsets."""
    v = torch.from_numpy(g["lap_v"]); t = torch.from_numpy(g["lap_t"]); off = torch.from_numpy(g["lap_off"]).requires_grad_(True)
    l = losses.laplacian_smooth_loss(v + off, t)
    l.backward()
    assert abs(float(l.detach()) - float(g["lap_loss"])) <= 2e-6 * abs(float(g["lap_loss"]))
    np.testing.assert_allclose(off.grad.numpy(), g["lap_goff"], rtol=1e-4, atol=1e-6 * float(np.abs(g["lap_goff"]).max()))
    with pytest.raises(NotImplementedError):
        losses.laplacian_smooth_loss(v, t, cotan=True)


def _cube():
    v = torch.tensor([[x, y, z] for x in (0., 1.) for y in (0., 1.) for z in (0., 1.)])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = torch.tensor([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))])
    return v, f


def test_pytorch3d_defined_regularisers_closed_form():
    """mesh_edge_loss / mesh_normal_consistency follow pytorch3d's published definitions (pytorch3d is not in the reference tree: unpinned).
    Unit square of two triangles: edges 1, 1, 1, 1, sqrt 2 -> mean squared length 6 / 5; coplanar faces -> consistency 0.  Unit cube of twelve
    triangles: 18 edges (12 of length 1, 6 face diagonals) -> (12 + 12) / 18; 6 coplanar pairs + 12 right-angle pairs -> 12 / 18.  Regular
    tetrahedron: outward normals meet at cos = -1/3 -> 4 / 3.  Orientation of the index order must not matter."""
    sq_v = torch.tensor([[0., 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]); sq_f = torch.tensor([[0, 1, 2], [0, 2, 3]])
    assert abs(float(losses.mesh_edge_loss(sq_v, sq_f)) - 1.2) < 1e-6
    assert abs(float(losses.mesh_normal_consistency(sq_v, sq_f))) < 1e-6
    assert abs(float(losses.mesh_normal_consistency(sq_v, torch.tensor([[0, 1, 2], [2, 0, 3]])))) < 1e-6
    v, f = _cube()
    assert abs(float(losses.mesh_edge_loss(v, f)) - 24.0 / 18.0) < 1e-6
    assert abs(float(losses.mesh_normal_consistency(v, f)) - 12.0 / 18.0) < 1e-6
    assert abs(float(losses.mesh_edge_loss(v, f, target_length=1.0)) - 6 * (2 ** 0.5 - 1) ** 2 / 18) < 1e-6
    tv = torch.tensor([[1., 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]]); tf = torch.tensor([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]])
    assert abs(float(losses.mesh_normal_consistency(tv, tf)) - 4.0 / 3.0) < 1e-6
    assert float(losses.mesh_normal_consistency(sq_v, sq_f[:1])) == 0.0          # no shared edge: zero, still differentiable


def test_stage1_loss_is_the_sum_of_its_terms_and_step_rescales():
    """stage1_loss assembles train_step's terms with main.py's default weights; stage1_optimizer_step applies x 64 to the environment-map gradient
    and / 8 to the hash-grid gradient before the material / light steps and clamps the light at 0.01 (nerf/utils.py:1565-1589) — seen through
    plain SGD, where the parameter change is the (rescaled) gradient."""
    torch.manual_seed(5)
    n = 96
    v, f = _cube()
    voff = (torch.rand(8, 3) * 0.1).requires_grad_(True)
    base = torch.full((4, 8, 3), 0.5).requires_grad_(True); grid = torch.rand(64, 2).requires_grad_(True)
    gt = torch.rand(n, 3); gt_lin = gt ** 2.2
    def outputs():
        lit = base.mean() * (grid.mean() + 1.0)
        return dict(image=torch.rand(n, 3, generator=torch.Generator().manual_seed(1)), image_brdf=(torch.rand(n, 3, generator=torch.Generator().manual_seed(2)) * lit).clamp(0, 1),
                    diffuse_light=torch.rand(n, 3, generator=torch.Generator().manual_seed(3)) * lit, specular_light=torch.rand(n, 3, generator=torch.Generator().manual_seed(4)) * lit * 0.2,
                    img_brdf_indirect=torch.full((n, 3), 0.05), kd_grad=torch.rand(n, 3, generator=torch.Generator().manual_seed(6)) * grid.mean(),
                    ks_grad=torch.rand(n, 1, generator=torch.Generator().manual_seed(7)), normal_grad=torch.rand(n, 1, generator=torch.Generator().manual_seed(8)))
    opt = types.SimpleNamespace(use_brdf=True)
    o = outputs()
    total = losses.stage1_loss(o, gt, gt_lin, opt, vertices=v, voffsets=voff, triangles=f)
    parts = (((o["image"] - gt) ** 2).mean(-1) + 0.02 * (o["image_brdf"] - gt).abs().mean(-1)).mean() \
        + losses.shading_loss(o["diffuse_light"], o["specular_light"], gt_lin - o["img_brdf_indirect"], 0.0015, 0.000025) \
        + losses.material_smoothness_grad(o["kd_grad"], o["ks_grad"], o["normal_grad"], 0.005, 0.0025, 0.00025) \
        + 0.001 * losses.laplacian_smooth_loss(v + voff, f) + 0.1 * losses.offsets_loss(voff)
    assert abs(float(total.detach()) - float(parts.detach())) < 1e-7
    # the outer mesh of --bound > 1 counts a tenth
    assert abs(float(losses.offsets_loss(voff, 5).detach()) - float(((voff[:5] ** 2).sum(-1).mean() + 0.1 * (voff[5:] ** 2).sum(-1).mean()).detach())) < 1e-7
    gb, gg, gv = torch.autograd.grad(total, (base, grid, voff))
    o_geo = torch.optim.SGD([voff], lr=1.0); o_mat = torch.optim.SGD([grid], lr=1.0); o_light = torch.optim.SGD([base], lr=1.0)
    b0, g0, v0 = base.detach().clone(), grid.detach().clone(), voff.detach().clone()
    val = losses.stage1_optimizer_step(losses.stage1_loss(outputs(), gt, gt_lin, opt, vertices=v, voffsets=voff, triangles=f), o_geo, o_mat, o_light,
                                       light_base=base, encoder_params=grid)
    assert abs(val - float(total.detach())) < 1e-7
    np.testing.assert_allclose((v0 - voff.detach()).numpy(), gv.numpy(), rtol=1e-5, atol=2e-8)      # v0 - (v0 - g) rounds at the scale of v0
    np.testing.assert_allclose((g0 - grid.detach()).numpy(), gg.numpy() / 8.0, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose((b0 - base.detach()).numpy(), gb.numpy() * 64.0, rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        base.fill_(0.0)
    losses.stage1_optimizer_step(losses.stage1_loss(outputs(), gt, gt_lin, opt), o_geo, o_mat, o_light, light_base=base, encoder_params=grid)
    assert float(base.detach().min()) >= np.float32(0.01)


def test_srgb_to_linear_matches_the_reference_function(g):
    """harness.srgb_to_linear against nerf/utils.py:57-58 run on the same values (the training images' linearisation for the shading loss)."""
    import torch
    from mirres_restir_nerf_mesh_amd import harness
    got = harness.srgb_to_linear(torch.from_numpy(g["s2l_in"])).numpy()
    assert np.array_equal(got, g["s2l_out"])
    back = harness.linear2srgb(torch.from_numpy(g["s2l_out"])).numpy()                      # and the forward curve inverts it (up to its 1e-6 offset)
    assert np.abs(back - g["s2l_in"]).max() < 2e-5


def test_projection_matrix_matches_the_reference_dataset(g):
    """harness.mvp_from_pose with an identity pose against the projection matrix nerf/provider.py:277-287 builds (its own statements, executed)."""
    import torch
    from mirres_restir_nerf_mesh_amd import harness
    H, W = (int(v) for v in g["proj_hw"]); fl = float(g["proj_fl"])
    got = harness.mvp_from_pose(torch.eye(4), (fl, fl, W * 0.5, H * 0.5), H, W, near=float(g["proj_near"])).numpy()
    np.testing.assert_allclose(got, g["proj_out"], rtol=1e-6, atol=1e-7)


def test_stage1_loss_matches_the_reference_train_step(g):
    """losses.stage1_loss — and the target preparation of scripts/train_stage1.py (white background compositing, linearised target) — against the
    reference's own Trainer.train_step (nerf/utils.py:912-1135, the method executed from its AST with render_stage1 replaced by prepared outputs;
    gen_reference_losses.py): the scalar and its gradients w.r.t. every rendered output and the vertex offsets, with main.py's default weights."""
    import types
    import torch
    from mirres_restir_nerf_mesh_amd import harness, losses
    rgba = torch.from_numpy(g["ts_rgba"])
    gt = rgba[:, :3] * rgba[:, 3:] + (1 - rgba[:, 3:])                                  # :948-955 with --background white
    gt_lin = harness.srgb_to_linear(rgba[:, :3]) * rgba[:, 3:]
    np.testing.assert_allclose(gt.numpy(), g["ts_gt"], rtol=0, atol=1e-7)
    names = ("image", "image_brdf", "diffuse_light", "specular_light", "kd_grad", "ks_grad", "normal_grad")
    outs = {k: torch.from_numpy(g["ts_in_" + k]).clone().requires_grad_(True) for k in names}
    voff = torch.from_numpy(g["ts_voff"]).clone().requires_grad_(True)
    verts = torch.from_numpy(g["lap_v"]); faces = torch.from_numpy(g["lap_t"])
    loss = losses.stage1_loss(dict(outs, img_brdf_indirect=torch.from_numpy(g["ts_indirect"])), gt, gt_lin, types.SimpleNamespace(use_brdf=True), vertices=verts, voffsets=voff,
                              triangles=faces)
    assert abs(float(loss.detach()) - float(g["ts_loss"])) <= 1e-6 * abs(float(g["ts_loss"]))
    loss.backward()
    for k in names:
        ref = g["ts_g_" + k]
        np.testing.assert_allclose(outs[k].grad.numpy(), ref, rtol=1e-5, atol=1e-6 * max(np.abs(ref).max(), 1e-12), err_msg=k)
    np.testing.assert_allclose(voff.grad.numpy(), g["ts_gvoff"], rtol=1e-5, atol=1e-6 * np.abs(g["ts_gvoff"]).max())


def test_optimizer_step_matches_the_reference_epoch_loop(g):
    """losses.stage1_optimizer_step against the reference's own Trainer.train_one_epoch (nerf/utils.py:1518-1660, the method executed from its AST with
    the reference's EnvironmentLight class; gen_reference_losses.py): three iterations from the same initial state on the same closed-form loss — geometry
    step and schedule, light gradient x 64, hash-grid gradient / 8, material and light steps and schedules, the clamp of the light at 0.01 — must leave
    every parameter where the reference's loop left it."""
    import torch
    from mirres_restir_nerf_mesh_amd import losses
    t = lambda k: torch.from_numpy(g[k]).clone()
    voff = t("os_voff0").requires_grad_(True); grid = t("os_grid0").requires_grad_(True); w = t("os_w0").requires_grad_(True); light = t("os_light0").requires_grad_(True)
    cv, cg, cw, cl = t("os_c_voff"), t("os_c_grid"), t("os_c_w"), t("os_c_light")
    f = lambda: (cv * voff).sum() + 3.0 * (voff ** 2).sum() + (cg * grid).sum() + (cw * w).sum() + 0.5 * (w ** 2).sum() + (cl * light).sum() + 0.2 * (light ** 2).sum()
    iters = 7500
    o_geo = torch.optim.Adam([{"params": [voff], "lr": 1e-4, "weight_decay": 0}], eps=1e-15)
    o_mat = torch.optim.Adam([{"params": [grid, w], "lr": 0.03}]); o_lgt = torch.optim.Adam([{"params": [light], "lr": 0.09}])
    s_geo = torch.optim.lr_scheduler.LambdaLR(o_geo, lambda it: 0.01 + 0.99 * (it / 500) if it <= 500 else 0.1 ** ((it - 500) / (iters - 500)))
    brdf = lambda it: max(0.0, 10 ** (-it * 0.0002))
    s_mat = torch.optim.lr_scheduler.LambdaLR(o_mat, brdf); s_lgt = torch.optim.lr_scheduler.LambdaLR(o_lgt, brdf)
    vals = []
    for _ in range(int(g["os_steps"])):
        for o in (o_geo, o_mat, o_lgt):
            o.zero_grad()
        vals.append(losses.stage1_optimizer_step(f(), o_geo, o_mat, o_lgt, light_base=light, encoder_params=grid, scheduler=s_geo, scheduler_mat=s_mat, scheduler_light=s_lgt))
    assert abs(np.mean(vals) - float(g["os_losses"][0])) <= 1e-6 * abs(float(g["os_losses"][0]))
    for name, p in (("voff", voff), ("grid", grid), ("w", w), ("light", light)):
        np.testing.assert_allclose(p.detach().numpy(), g["os_%s3" % name], rtol=1e-6, atol=1e-9, err_msg=name)
    assert float(light.detach().min()) == np.float32(0.01)                    # the clamp was active in this run
