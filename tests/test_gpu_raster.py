"""GPU: the G-buffer front end — raster record by ray casting, dr.interpolate's contract forward and backward (against a plain torch
fp32 restatement of the formula), auto_normals against the reference's own function (tests/golden/ref_python.npz)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_raster_record_and_interpolation(oracle, scene_mod):
    import torch
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR, raster as RS
    v, t = scene_mod.make_mesh(4, 8)
    W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
    eye, rd = scene_mod.camera_rays(64, 48)
    n = rd.shape[0]
    o = torch.from_numpy(np.repeat(eye[None], n, 0)).cuda(); d = torch.from_numpy(rd).cuda()
    rast = RS.rasterize_raycast(W, o, d)
    ref = W.trace(o, d, closest=True)
    hit = ref["hit"] > 0
    assert torch.equal(rast[:, 3] > 0, hit) and torch.equal((rast[:, 3].long() - 1)[hit], ref["prim"].long()[hit]) and torch.equal(rast[:, 2][hit], ref["t"][hit])
    assert (rast[~hit] == 0).all() and 0.2 < float(hit.float().mean()) < 0.9
    u, vv = rast[hit, 0], rast[hit, 1]
    assert float(u.min()) >= -1e-5 and float(vv.min()) >= -1e-5 and float((u + vv).max()) <= 1 + 1e-5
    # interpolating the vertex positions reproduces the hit points
    vt = torch.from_numpy(v).cuda().requires_grad_(True); tt = torch.from_numpy(t).cuda()
    xyz = RS.interpolate(vt, rast, tt)
    assert float((xyz.detach()[hit] - ref["pos"][hit]).abs().max()) < 2e-5 and (xyz[~hit] == 0).all()
    # forward / backward against the plain torch formula, 5 channels, gradients to the attributes and to (u, v)
    attr = torch.randn((v.shape[0], 5), device="cuda", requires_grad=True)
    rast_g = rast.clone().requires_grad_(True)
    out = RS.interpolate(attr, rast_g, tt)
    idx = (rast[:, 3].long() - 1).clamp(min=0); tri = tt.long()[idx]
    attr2 = attr.detach().clone().requires_grad_(True); rast2 = rast.detach().clone().requires_grad_(True)
    b0, b1 = rast2[:, 0:1], rast2[:, 1:2]
    ref_out = (b0 * attr2[tri[:, 0]] + b1 * attr2[tri[:, 1]] + (1 - b0 - b1) * attr2[tri[:, 2]]) * hit[:, None]
    assert torch.allclose(out, ref_out, rtol=1e-6, atol=1e-6)
    gw = torch.randn_like(out)
    (out * gw).sum().backward(); (ref_out * gw).sum().backward()
    assert torch.allclose(attr.grad, attr2.grad, rtol=1e-4, atol=1e-4)            # atomics: summation order differs
    assert torch.allclose(rast_g.grad[:, 0:2], rast2.grad[:, 0:2], rtol=1e-5, atol=1e-5) and (rast_g.grad[:, 2:] == 0).all()
    # gradient of a loss on the interpolated positions reaches the vertices of the visible triangles only
    (xyz ** 2).sum().backward()
    seen = torch.zeros(v.shape[0], dtype=torch.bool, device="cuda"); seen[tt.long()[(rast[:, 3].long() - 1)[hit]].reshape(-1)] = True
    assert (vt.grad[~seen] == 0).all() and float(vt.grad[seen].abs().sum()) > 0
    assert RS.rasterize_raycast(W, o[:0], d[:0]).shape == (0, 4)


def test_auto_normals_matches_the_reference(scene_mod):
    import torch
    from mirres_restir_nerf_mesh_amd import raster as RS
    g = np.load(os.path.join(HERE, "golden", "ref_python.npz"))
    vn, idx = RS.auto_normals(torch.from_numpy(g["an_vert"]).cuda(), torch.from_numpy(g["an_tri"]).cuda())
    np.testing.assert_allclose(vn.cpu().numpy(), g["an_out"], rtol=2e-5, atol=2e-6)
    assert np.array_equal(idx.cpu().numpy(), g["an_tri"])


def test_stage1_front_half_gbuffer(scene_mod):
    """harness.build_gbuffer_stage1 = render_stage1's front half on the engine's operators: agrees with the face-normal G-buffer on visibility and
    position, produces unit shading normals close to the geometric ones on the smooth mesh, and carries gradients back to the vertex positions."""
    import torch
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
    v, t = scene_mod.make_mesh(5, 8)
    vt = torch.from_numpy(v).cuda().requires_grad_(True); tt = torch.from_numpy(t).cuda()
    W = RR.restirbvhWorker(vt.detach(), tt); W.update_mesh(W.vrt, W.v_ind)
    g1 = harness.build_gbuffer_stage1(W, vt, tt, 48, 40)
    g0 = harness.build_gbuffer(W, 48, 40)
    fg = g0["occ"][:, 0] > 0.5
    assert torch.equal(g1["occ"], g0["occ"]) and float((g1["pos"].detach()[fg] - g0["pos"].detach()[fg]).abs().max()) < 2e-5
    n1 = g1["normal"][fg]
    assert torch.allclose(n1.norm(dim=1), torch.ones_like(n1[:, 0]), atol=1e-5)
    cosang = (n1 * g0["normal"][fg]).sum(1)
    assert float(cosang.mean()) > 0.97 and float((cosang > 0.5).float().mean()) > 0.99        # smooth normals stay near the face normals of the smooth mesh
    assert (g1["normal"][~fg] == 0).all()
    loss = (g1["pos"] ** 2).sum() + (g1["normal"] * torch.tensor([0.3, 0.5, 0.8], device="cuda")).sum()
    loss.backward()
    assert torch.isfinite(vt.grad).all() and float(vt.grad.abs().sum()) > 0
    out = RR.render_fused(__import__("mirres_restir_nerf_mesh_amd._ops", fromlist=["get_ctx"]).get_ctx(g1["fx"], g1["fy"]), W, None, False, (1, 1, 1),
                          torch.from_numpy(scene_mod.make_env(32, 64)).cuda(), g1["occ"].clone(), g1["normal"].detach().contiguous(), g1["depth"].detach().contiguous(), g1["kd"],
                          g1["rm"], g1["ray_dir"], g1["pos"].detach().contiguous(), 2, 2, 2, 2.0, 0.1, 0.001, 5)[0]
    assert torch.isfinite(out[0]).all() and float(out[0][fg].mean()) > 0.01                    # and it feeds the path


def test_texture_taps_against_grid_sample():
    """raster.texture (dr.texture, linear filter, clamp boundary) against torch's grid_sample with the same convention (bilinear, border padding,
    align_corners False), forward and the gradient to the texture; coordinates inside, on the border and outside [0, 1]."""
    import torch
    from mirres_restir_nerf_mesh_amd import raster as RS
    gen = torch.Generator(device="cuda").manual_seed(4)
    H, W, C = 13, 21, 5
    tex = torch.rand((H, W, C), device="cuda", generator=gen, requires_grad=True)
    uv = torch.rand((4000, 2), device="cuda", generator=gen) * 1.3 - 0.15
    uv[:4] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.5 / W, 0.5 / H], [1.0 - 0.5 / W, 0.3]], device="cuda")
    out = RS.texture(tex, uv)
    tex2 = tex.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.grid_sample(tex2.permute(2, 0, 1)[None], (uv * 2 - 1).view(1, 1, -1, 2), mode="bilinear", padding_mode="border", align_corners=False)[0, :, 0].t()
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
    gw = torch.randn_like(out)
    (out * gw).sum().backward(); (ref * gw).sum().backward()
    assert torch.allclose(tex.grad, tex2.grad, rtol=1e-4, atol=1e-4)


def test_pose_camera_gbuffer_and_test_view(scene_mod):
    """harness.build_gbuffer_from_pose (dataset camera: cam2world pose + (fx, fy, cx, cy), rays as nerf/utils.py:get_rays forms them) sees what
    build_gbuffer sees for the same camera, hands the path unit shading directions, and harness.test_view (the `--test --spp N` frame of the BRDF
    branch: G-buffer -> mirres_render -> tone curve, alpha, SSAA, white background) returns a finite [H, W, 3] image in [0, 1] whose background is
    exactly white and whose ssaa 2 rendering agrees with the ssaa 1 one up to sampling noise and edge coverage."""
    import torch
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
    v, t = scene_mod.make_mesh(4, 8)
    W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
    H, Wd = 40, 48
    az = el = np.deg2rad(30.0)
    eye = 3.2 * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    fwd = -eye / np.linalg.norm(eye); right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right); cup = np.cross(right, fwd)
    pose = np.eye(4); pose[:3, :3] = np.stack([right, cup, -fwd], axis=1); pose[:3, 3] = eye
    focal = 0.5 * Wd / np.tan(0.5 * 0.6911)
    intr = (focal, focal, Wd * 0.5, H * 0.5)
    pose_t = torch.from_numpy(pose.astype(np.float32))
    g0 = harness.build_gbuffer(W, H, Wd)
    g1 = harness.build_gbuffer_from_pose(W, pose_t, intr, H, Wd)
    both = (g0["occ"][:, 0] > 0.5) & (g1["occ"][:, 0] > 0.5)
    assert int(((g0["occ"] > 0.5) != (g1["occ"] > 0.5)).sum()) <= 2 and int(both.sum()) > 200
    assert float((g0["pos"][both] - g1["pos"][both]).abs().max()) < 1e-4
    d0 = g0["ray_dir"] / g0["ray_dir"].norm(dim=1, keepdim=True)
    assert float((d0 - g1["ray_dir"]).abs().max()) < 1e-5 and torch.allclose(g1["ray_dir"].norm(dim=1), torch.ones(H * Wd, device="cuda"), atol=1e-5)
    env = torch.from_numpy(scene_mod.make_env(32, 64)).cuda()
    img2 = harness.test_view(W, None, env, pose_t, intr, H, Wd, spp=8, ssaa=2, random_offset=5)
    img1 = harness.test_view(W, None, env, pose_t, intr, H, Wd, spp=8, ssaa=1, random_offset=5)
    for img in (img1, img2):
        assert tuple(img.shape) == (H, Wd, 3) and torch.isfinite(img).all() and float(img.min()) >= 0.0 and float(img.max()) <= 1.0 + 1e-6      # linear2srgb_torch(1) = 1.055 (1 + 1e-6)^(1/2.4) - 0.055, a hair above 1
        assert torch.equal(img[0, 0], torch.ones(3, device="cuda")) and torch.equal(img[-1, -1], torch.ones(3, device="cuda"))
    bg = (g1["occ"].view(H, Wd) < 0.5)
    assert torch.equal(img1[bg], torch.ones_like(img1[bg]))
    assert harness.psnr(img2, img1) > 18.0


def test_evaluation_script_relighting_and_sharded_views(tmp_path):
    """scripts/evaluate.py end to end on a synthetic workspace (BASELINE configs[3]'s shape in small): the trained map, then relighting with an external
    Radiance .hdr map of another size and albedo scaling (--envmap_path / --albedo_scale_*), then the same relit job on two ranks (gloo, both on this
    GPU) with the exact strip sharding — whose PNG files must equal the single-process ones byte for byte."""
    import os, subprocess, sys
    import torch
    from mirres_restir_nerf_mesh_amd import harness, scene
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = str(tmp_path / "sky.hdr")
    harness.write_hdr(hdr, scene.make_env(48, 96, sun=40.0) * np.array([1.0, 0.8, 0.6], np.float32))
    base = [sys.executable, os.path.join(root, "scripts", "evaluate.py"), "--synthetic", "--H", "64", "--W", "64", "--spp", "4", "--ssaa", "2", "--limit", "2"]
    env = dict(os.environ, MIRRES_DIST_BACKEND="gloo")
    def run(ws, extra, launcher=()):
        cmd = list(launcher) + base[(1 if launcher else 0):] + ["--workspace", str(tmp_path / ws)] + extra
        if launcher:
            cmd = [sys.executable] + cmd
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        d = tmp_path / ws / "results_brdf"
        return {f: open(d / f, "rb").read() for f in sorted(os.listdir(d)) if f.endswith(".png")}, r.stdout
    plain, _ = run("a", [])
    relit, out = run("b", ["--envmap_path", hdr, "--albedo_scale_x", "0.9", "--albedo_scale_y", "0.8", "--albedo_scale_z", "0.7"])
    assert len(plain) >= 2 and plain.keys() == relit.keys() and all(plain[k] != relit[k] for k in plain) and "rendered 2 views" in out
    launcher = ("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(36000 + os.getpid() % 2000))
    strips, out2 = run("c", ["--envmap_path", hdr, "--albedo_scale_x", "0.9", "--albedo_scale_y", "0.8", "--albedo_scale_z", "0.7", "--shard", "strips"], launcher)
    assert strips == relit and "on 2 GPU(s) [strips]" in out2
    views, out3 = run("d", ["--envmap_path", hdr, "--albedo_scale_x", "0.9", "--albedo_scale_y", "0.8", "--albedo_scale_z", "0.7", "--shard", "views"], launcher)
    assert views == relit and "[views]" in out3
