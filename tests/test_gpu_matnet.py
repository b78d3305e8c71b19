"""GPU parity of the material field: hash-grid encoding BIT-EXACT (fp16), the MLP BIT-EXACT too — per-lane fp32 kernel and the fp32-MFMA kernel
(v_mfma_f32_32x32x2_f32 = an fmaf chain in k order) against the oracle's fmaf chain, sigmoid through the shared include/mirres_fmath.h."""
import numpy as np
import pytest

from util import same_bits

pytestmark = pytest.mark.gpu


def test_matnet_forward(oracle, scene_mod):
    import ctypes as C
    import torch
    from mirres_restir_nerf_mesh_amd import _lib
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    params, w0, w1, w2 = scene_mod.make_matnet_params(seed=1)
    mn, mx = scene_mod.material_min_max()
    keep = oracle.Keep()
    om = oracle.matnet_struct(keep, params, w0, w1, w2, (-1, -1, -1), (1, 1, 1), mn, mx)
    rng = np.random.default_rng(2)
    pos = (rng.random((20000, 3)) * 2.4 - 1.2).astype(np.float32)   # includes points outside the AABB (clamped)
    pos[:8] = [[-1, -1, -1], [1, 1, 1], [0, 0, 0], [1, -1, 0.5], [0.999999, 0.3, -0.7], [-1.5, 2, 0], [0.25, 0.25, 0.25], [1e-8, -1e-8, 0]]
    ref_out = oracle.matnet(om, pos)
    x01 = np.clip((pos + 1) / 2, 0, 1).astype(np.float32)
    ref_enc = oracle.hashgrid_encode(om, np.clip((pos - np.float32(-1)) / (np.float32(1) - np.float32(-1)), 0, 1).astype(np.float32))
    dp = torch.from_numpy(params).cuda(); g16 = torch.empty(params.size, dtype=torch.int16, device="cuda")
    check(lib().mirres_matnet_pack_grid(dp.data_ptr(), g16.data_ptr(), params.size, None), "pack")
    assert np.array_equal(g16.cpu().numpy().view(np.uint16), oracle.to_f16_bits(params))       # RNE fp32->fp16
    st = _lib.MatNet(); t = [torch.from_numpy(a).cuda() for a in (w0, w1, w2)]
    st.grid_f16 = g16.data_ptr(); st.w0, st.w1, st.w2 = (a.data_ptr() for a in t)
    st.aabb_min[:] = [-1, -1, -1]; st.aabb_max[:] = [1, 1, 1]; st.out_min[:] = mn.tolist(); st.out_max[:] = mx.tolist()
    n = len(pos); dpos = torch.from_numpy(pos).cuda(); out = torch.empty((n, 6), device="cuda"); enc = torch.empty((n, 32), dtype=torch.int16, device="cuda")
    check(lib().mirres_matnet_fwd(C.byref(st), dpos.data_ptr(), n, out.data_ptr(), enc.data_ptr(), None), "fwd")
    assert np.array_equal(enc.cpu().numpy().view(np.uint16), ref_enc)
    same_bits(out.cpu().numpy(), ref_out, "per-lane material field")
    assert (out[:, 3] == 0).all()                                   # channel 3 is constant 0 (min = max = 0)
    assert lib().mirres_matnet_grid_entries() == 6299960
    # dense levels agree with a direct trilinear lookup: independent check of the index arithmetic (level 0: res 16, 4096 entries)
    tab = oracle.to_f16_bits(params[:8192]).view(np.float16).astype(np.float64).reshape(16, 16, 16, 2)  # [z][y][x]
    p = x01[100:400].astype(np.float64) * 15.0 + 0.5
    i0 = np.floor(p).astype(int); fr = p - i0
    acc = np.zeros((300, 2))
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                wgt = (fr[:, 0] if dx else 1 - fr[:, 0]) * (fr[:, 1] if dy else 1 - fr[:, 1]) * (fr[:, 2] if dz else 1 - fr[:, 2])
                ix, iy, iz = (i0[:, 0] + dx), (i0[:, 1] + dy), (i0[:, 2] + dz)
                idx = (ix + iy * 16 + iz * 256) % 4096
                acc += wgt[:, None] * tab.reshape(4096, 2)[idx]
    got = ref_enc[100:400, :2].view(np.float16).astype(np.float64)
    np.testing.assert_allclose(got, acc, rtol=0, atol=3e-3 * np.abs(acc).max() + 1e-4)


def test_mfma_mlp_matches_fp32_chain(oracle, scene_mod):
    """The MFMA-tiled MLP (fp32 matrix pipe, sixteen K = 2 steps per layer in ascending k) IS the fp32 fmaf-chain MLP: bit-equal to the per-lane kernel
    and to the oracle, for every ragged tile / block tail."""
    import torch
    from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
    mn, mx = scene_mod.material_min_max(me_max=0.7)
    mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=9)
    with torch.no_grad():
        mlp.encoder.params.mul_(3e3)
    g = torch.Generator(device="cuda").manual_seed(0)
    for n in (1, 63, 64, 257, 5000, 70001):        # ragged tails of the 64-point MFMA tiles / 256-point blocks
        pts = torch.rand((n, 3), device="cuda", generator=g) * 2 - 1
        ref = mlp.sample_no_di(pts)                 # per-lane fp32 kernel (itself checked against the oracle above)
        enc = mlp.encode(pts)
        got = mlp.mlp_on_encoding(enc)
        assert got.shape == (n, 6)
        assert torch.equal(got, ref), (n, float((got - ref).abs().max()))
    keep = oracle.Keep()
    w = [mlp.net.net[i].weight.detach().cpu().numpy() for i in (0, 2, 4)]
    om = oracle.matnet_struct(keep, mlp.encoder.params.detach().cpu().numpy(), w[0], w[1], w[2], (-1, -1, -1), (1, 1, 1), mn, mx)
    pts = torch.rand((4096, 3), device="cuda", generator=g) * 2 - 1
    same_bits(mlp.mlp_on_encoding(mlp.encode(pts)).cpu().numpy(), oracle.matnet(om, pts.cpu().numpy()), "MFMA material MLP vs oracle")
    assert mlp.mlp_on_encoding(torch.empty((0, 32), dtype=torch.float16, device="cuda")).shape == (0, 6)


def test_material_field_matches_the_reference_classes(scene_mod):
    """MLPTexture3D.sample / sample_no_di of the package (hash grid + MFMA MLP through the C ABI) against the outputs of the REFERENCE's own
    MLPTexture3D / _MLP classes run over the oracle's hash-grid encoder (tests/golden/ref_loop.npz, gen_reference_loop.py): same seeded weights,
    points inside, outside (clamped) and on the corners of the AABB."""
    import os
    import torch
    from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_loop.npz"))
    params, w0, w1, w2 = scene_mod.make_matnet_params(seed=0)
    mn, mx = scene_mod.material_min_max()
    mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=1)
    with torch.no_grad():
        mlp.encoder.params.copy_(torch.from_numpy(params).cuda())
        for i, w in zip((0, 2, 4), (w0, w1, w2)):
            mlp.net.net[i].weight.copy_(torch.from_numpy(w).cuda())
    x = torch.from_numpy(g["mat_pts"]).cuda()
    np.testing.assert_allclose(mlp.sample_no_di(x).cpu().numpy(), g["mat_out"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(mlp.sample(x).detach().cpu().numpy(), g["mat_out"], rtol=0, atol=1e-5)


def test_checkpoint_round_trip_through_the_reference_layout(scene_mod, tmp_path):
    """checkpoint.save_checkpoint writes Trainer.save_checkpoint's layout (nerf/utils.py:1843-1854); read_checkpoint + apply_checkpoint restore a
    freshly constructed MLPTexture3D to bit-identical outputs, hand back offsets and light, and refuse a field of the wrong size."""
    import torch
    from mirres_restir_nerf_mesh_amd import checkpoint as CK
    from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
    mn, mx = scene_mod.material_min_max()
    mk = lambda seed: MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6,
                                   min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=seed)
    torch.manual_seed(1); a = mk(5)
    with torch.no_grad():
        a.encoder.params.mul_(1e3)
    voff = torch.rand(100, 3, device="cuda") * 0.01; light = torch.rand(16, 32, 3, device="cuda") + 0.01
    p = str(tmp_path / "ngp_stage1.pth")
    CK.save_checkpoint(p, a, voff, light, epoch=3, global_step=77)
    torch.manual_seed(2); b = mk(9)
    x = torch.rand(5000, 3, device="cuda") * 2 - 1
    assert not torch.equal(a.sample(x), b.sample(x))
    r = CK.read_checkpoint(p)
    voff2, light2 = CK.apply_checkpoint(r, b, n_vertices=100)
    assert torch.equal(a.sample(x), b.sample(x)) and torch.equal(voff2, voff) and torch.equal(light2, light) and r["global_step"] == 77
    # the constants that decode the field (AABB / output ranges: the reference's --bound, --roughness_min, --me_max) are recorded from the module
    mc = r["material_config"]
    assert mc["bound"] == 1.0 and mc["roughness_min"] == pytest.approx(0.08) and mc["me_max"] == 0.0
    aabb, mn2, mx2 = CK.material_field_args(CK.resolve_material_config(mc))
    assert aabb.tolist() == [-1, -1, -1, 1, 1, 1] and np.allclose(mn2.numpy(), mn) and np.allclose(mx2.numpy(), mx)
    r["grid_params"] = r["grid_params"][:-2]
    with pytest.raises(ValueError, match="encoder.params"):
        CK.apply_checkpoint(r, b)
