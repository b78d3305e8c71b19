"""CPU: host-side logic — synthetic scene generator, camera model, sharding arithmetic, post-processing formulas, and the N>1 exchange
(torch.distributed gloo, world_size 2) of mirres-restir_nerf_mesh_amd/dist.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import spawn_ranks  # noqa: E402


def test_scene_is_deterministic_and_well_formed(scene_mod):
    v, t = scene_mod.make_mesh(3, 8)
    v2, t2 = scene_mod.make_mesh(3, 8)
    assert np.array_equal(v, v2) and np.array_equal(t, t2)
    assert v.dtype == np.float32 and t.dtype == np.int32 and t.min() == 0 and t.max() == len(v) - 1
    assert len(t) == 20 * 4 ** 3 + 2 * 64 and np.abs(v).max() < 1.0                     # inside [-1,1]^3 (--bound 1)
    tv = v[t]
    ext = tv.max(1) - tv.min(1)
    assert ext.min() > 1e-5, "no axis-aligned (zero-thickness) triangle: the reference never enters such boxes (helperDi.slang:165)"
    vfull, tfull = scene_mod.make_mesh(7, 64)
    assert len(tfull) == 327680 + 8192


def test_camera_model(scene_mod):
    eye, rd = scene_mod.camera_rays(8, 10, azimuth_deg=30, elevation_deg=30, radius=3.2)
    assert rd.shape == (80, 3) and abs(np.linalg.norm(eye) - 3.2) < 1e-5
    c = rd.reshape(8, 10, 3)
    centre = (c[3, 4] + c[3, 5] + c[4, 4] + c[4, 5]) / 4
    np.testing.assert_allclose(centre / np.linalg.norm(centre), -eye / np.linalg.norm(eye), atol=1e-6)     # looks at the origin
    # NeRF-blender convention (nerf/utils.py:408-416): +x right, +y up, camera looks down -z
    focal = 0.5 * 10 / np.tan(0.5 * 0.6911)
    assert abs(np.linalg.norm(c[0, 1] - c[0, 0]) - 1 / focal) < 1e-6
    assert c[0, 0, 2] > c[7, 0, 2]                                                         # image rows go down in world z


def test_spp_slices_partition_the_range():
    from mirres_restir_nerf_mesh_amd.dist import spp_slice
    for spp in (1, 2, 7, 128, 512):
        for world in (1, 2, 3, 4, 8):
            sl = [spp_slice(spp, r, world) for r in range(world)]
            assert sl[0][0] == 0 and sl[-1][1] == spp
            assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            sizes = [e - b for b, e in sl]
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == spp


def test_postprocess_formulas():
    import torch
    from mirres_restir_nerf_mesh_amd import harness
    x = torch.tensor([0.0, 0.001, 0.0031308, 0.2, 1.0])
    y = harness.linear2srgb(x)
    ref = np.where(x.numpy() <= 0.0031308, 12.92 * x.numpy(), 1.055 * (x.numpy() + 1e-6) ** (1 / 2.4) - 0.055)
    np.testing.assert_allclose(y.numpy(), ref, rtol=1e-6)
    a = torch.rand(4, 4, 3); b = a + 0.1
    assert abs(harness.psnr(a, b) - 20.0) < 1e-4                                          # -10 log10(0.01)


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirres_restir_nerf_mesh_amd.dist import allreduce_sums, spp_slice
    g = torch.Generator().manual_seed(100 + rank)
    sums = [torch.rand(50, 3, generator=g) * (spp_slice(10, rank, world)[1] - spp_slice(10, rank, world)[0]) for _ in range(6)]
    red = allreduce_sums([s.clone() for s in sums])
    # foreground-only exchange: background rows are zero on every rank (here: rows 10..29) and stay exactly zero; the others get the same sums
    occ = torch.ones(50); occ[10:30] = 0
    masked = [s.clone() * occ[:, None] for s in sums]
    red_fg = allreduce_sums([m.clone() for m in masked], occ=occ)
    torch.save(dict(local=sums, red=red, masked=masked, red_fg=red_fg), os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_allreduce_exchange_gloo_world2(tmp_path):
    """The only data-path collective of the sharded render: one all-reduce(sum) of the six accumulators."""
    import torch
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    spawn_ranks(_worker, (2, port, str(tmp_path)), 2)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt")); r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    for k in range(6):
        expect = r0["local"][k] + r1["local"][k]
        assert torch.allclose(r0["red"][k], expect) and torch.allclose(r1["red"][k], expect)
        assert r0["red"][k].shape == (50, 3)
        want = r0["masked"][k] + r1["masked"][k]
        assert torch.equal(r0["red_fg"][k], want) and torch.equal(r1["red_fg"][k], want) and float(r0["red_fg"][k][10:30].abs().max()) == 0.0


def test_strip_partition_and_halo_plan():
    """Strip sharding host logic: strips tile the frame, halos are 30 rows clipped at the image border, and the exchange plan of adjacent
    ranks is symmetric (what rank r sends to r+1 is exactly what r+1 expects in its top halo)."""
    from mirres_restir_nerf_mesh_amd import dist as D
    for fy, world in ((1600, 8), (1600, 3), (96, 2), (1601, 4)):
        rows = [D.strip_rows(fy, r, world) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == fy and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        for r, (y0, y1, lo, hi) in enumerate(rows):
            assert lo == max(0, y0 - 30) and hi == min(fy, y1 + 30)
            for peer, (sa, sb), (ra, rb) in D.halo_plan(fy, 8, r, world):
                # the rows we send are our own rows; the rows we receive are halo rows; global rows match the peer's view
                assert y0 - lo <= sa < sb <= y1 - lo and (rb <= y0 - lo or ra >= y1 - lo)
                back = [p for p in D.halo_plan(fy, 8, peer, world) if p[0] == r][0]
                plo = rows[peer][2]
                assert (lo + sa, lo + sb) == (plo + back[2][0], plo + back[2][1])      # our send range == the peer's receive range, in global rows
                assert (lo + ra, lo + rb) == (plo + back[1][0], plo + back[1][1])
    # cost-balanced heights: a frame whose lower half is foreground gets taller strips at the top; every strip keeps the halo's minimum height
    import torch
    fy, fx = 400, 10
    occ = torch.zeros(fy, fx); occ[200:] = 1
    b = D.strip_bounds(fy, 4, occ.reshape(-1, 1), fx)
    assert b[0] == 0 and b[-1] == fy and all(y1 - y0 >= 30 for y0, y1 in zip(b, b[1:]))
    assert b[1] - b[0] > b[4] - b[3]                                                      # cheap rows -> taller strip
    cost = D.BG_WEIGHT * fx + (1 - D.BG_WEIGHT) * occ.sum(1)
    per = [float(cost[y0:y1].sum()) for y0, y1 in zip(b, b[1:])]
    assert max(per) < 1.35 * min(per)
    assert D.strip_bounds(fy, 4) == [0, 100, 200, 300, 400]
    with pytest.raises(ValueError):
        D.strip_rows(100, 0, 8)                                                            # 12 rows per rank < 30-row halo


def test_strip_balancer_converges_on_a_cost_the_static_model_does_not_know():
    """dist.StripBalancer: strips timed by a synthetic truth (per-row foreground cost that varies by a factor of nine over the image + a fixed cost per strip, which the
    multiplicative update does not model) — the slowest strip starts 40-50 % above the mean under the static model and is within 4 % after three updates, for 2, 4 and 8
    ranks; boundaries stay legal partitions throughout; a rank list of the wrong length or with a zero is ignored."""
    import torch
    from mirres_restir_nerf_mesh_amd import dist as D
    torch.manual_seed(0)
    fy, fx = 1600, 400
    y = torch.arange(fy, dtype=torch.float64)
    occ = (((y[:, None] - 900).abs() < 500).float().expand(fy, fx) * (torch.rand(fy, fx) < 0.6)).float()
    rowc = (occ > 0.5).double().sum(1) * (1.0 + 0.8 * torch.sin(y / 150)) * 4e-3
    for world in (2, 4, 8):
        B = D.StripBalancer(fy, world)
        ratios = []
        for it in range(6):
            b = B.bounds(fx, occ.reshape(-1, 1))
            assert b[0] == 0 and b[-1] == fy and all(y1 - y0 >= D.HALO_ROWS for y0, y1 in zip(b, b[1:]))
            t = [60.0 + float(rowc[b[r]:b[r + 1]].sum()) for r in range(world)]
            ratios.append(max(t) / (sum(t) / world))
            B.update(t)
        assert ratios[0] > 1.3 and max(ratios[3:]) < 1.04, ratios
        before = B.corr.clone()
        B.update(t[:-1]); B.update([0.0] * world)
        assert torch.equal(before, B.corr)
        assert len(B.history) == 6


def _times_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirres_restir_nerf_mesh_amd import dist as D
    out[rank] = D.gather_times(10.0 + rank, rank, world)
    dist.destroy_process_group()


def test_strip_times_are_gathered_in_rank_order_gloo_world2():
    import torch.multiprocessing as mp
    mgr = mp.Manager(); out = mgr.dict()
    spawn_ranks(_times_worker, (2, 29600 + (os.getpid() % 200), out), 2)
    assert out[0] == [10.0, 11.0] and out[1] == [10.0, 11.0]


def _coupled_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirres_restir_nerf_mesh_amd import dist as D
    fy, fx = 400, 8
    occ = torch.ones(fy * fx, 1)
    # truth the static model does not know: the lower half of the image costs three times the upper half per row. In a real multi-rank run every sample's halo
    # exchange makes the ranks wait for each other: the WALL time of both strips is the slower strip's, whatever the partition.
    rowc = torch.where(torch.arange(fy) < fy // 2, torch.tensor(1.0), torch.tensor(3.0)).double()
    B = D.StripBalancer(fy, world)
    log = []
    for it in range(5):
        B.exchange(rank, measured=None if it == 0 else last)
        b = B.bounds(fx, occ)
        busy = [float(rowc[b[r]:b[r + 1]].sum()) for r in range(world)]
        wall = max(busy)                                  # coupled by the per-sample exchange
        last = (busy[rank], wall, wall - busy[rank])      # (own busy, total, waited): what StripBalancer.busy_ms() derives from its events
        log.append((list(b), busy, B.history[-1][1] if B.history else None))
    # an all-background view: nothing to exchange, no zero-size collective, the sums come back unchanged
    sums = [torch.zeros(fy * fx, 3) for _ in range(6)]
    red = D.allreduce_sums([x.clone() for x in sums], occ=torch.zeros(fy * fx, 1), used=D.USED_SUMS)
    ok_empty = all(torch.equal(a, b_) for a, b_ in zip(red, sums)) and D.sum_over_ranks(torch.zeros(0)).numel() == 0
    out[rank] = (log, ok_empty)
    dist.destroy_process_group()


def test_strip_balancer_is_fed_own_busy_times_gloo_world2():
    """two ranks whose strips cost 1 : 3 per row and whose wall times are EQUAL (coupled by the per-sample exchange). The gathered times are the strips'
    own busy times — they differ on the first frame — and the boundary moves towards balance; with wall times the balancer would never move. Also: an all-background
    occupancy short-circuits the sum exchange (no zero-size collective)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager(); out = mgr.dict()
    spawn_ranks(_coupled_worker, (2, 29900 + (os.getpid() % 200), out), 2)
    (log0, e0), (log1, e1) = out[0], out[1]
    assert e0 and e1
    assert log0 == log1                                              # every rank derives the same boundaries from the same gathered times
    b_first, busy_first, _ = log0[0]
    assert b_first == [0, 200, 400] and busy_first == [200.0, 600.0]
    gathered = log0[1][2]
    assert gathered == [200.0, 600.0]                                # own busy times travelled, not the (equal) wall times
    ratios = [max(x[1]) / (sum(x[1]) / 2) for x in log0]
    assert ratios[0] == 1.5 and ratios[-1] < 1.05, ratios            # the boundary moved down into the expensive half
    assert log0[-1][0][1] > 250


def _strip_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirres_restir_nerf_mesh_amd import dist as D
    fy, fx = 100, 6
    y0, y1, lo, hi = D.strip_rows(fy, rank, world)
    # "global truth": record value = global row * 1000 + column * 10 + field; a rank starts with only its own rows filled in
    gy = torch.arange(fy, dtype=torch.float32)[:, None, None] * 1000 + torch.arange(fx, dtype=torch.float32)[None, :, None] * 10 + torch.arange(8, dtype=torch.float32)[None, None, :]
    rec = torch.full((hi - lo, fx, 8), -1.0)
    rec[y0 - lo:y1 - lo] = gy[y0:y1]
    D.exchange_halos(rec, D.halo_plan(fy, fx, rank, world))
    own = [gy[y0:y1, :, :3].reshape(-1, 3).clone() * (k + 1) for k in range(6)]
    full = D.gather_rows(own, fy, fx, world)
    torch.save(dict(rec=rec, lo=lo, hi=hi, full=full), os.path.join(out, "s%d.pt" % rank))
    dist.destroy_process_group()


def test_strip_exchange_and_gather_gloo_world2(tmp_path):
    """The two data-path exchanges of the exact multi-GPU scheme on CPU tensors: after one halo exchange every rank's local frame (own +
    halo rows) equals the global records, and the row all-gather reassembles the full-frame sums on every rank."""
    import torch
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    spawn_ranks(_strip_worker, (2, port, str(tmp_path)), 2)
    fy, fx = 100, 6
    gy = torch.arange(fy, dtype=torch.float32)[:, None, None] * 1000 + torch.arange(fx, dtype=torch.float32)[None, :, None] * 10 + torch.arange(8, dtype=torch.float32)[None, None, :]
    for r in range(2):
        d = torch.load(os.path.join(tmp_path, "s%d.pt" % r))
        assert torch.equal(d["rec"], gy[d["lo"]:d["hi"]])
        for k in range(6):
            assert torch.equal(d["full"][k], gy[:, :, :3].reshape(-1, 3) * (k + 1))


def _grad_worker(rank, world, port, out, algo):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mirres_restir_nerf_mesh_amd.dist import allreduce_gradients
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.zeros(7, 3)); e = torch.nn.Parameter(torch.zeros(4, 5, 3)); unused = torch.nn.Parameter(torch.zeros(2))
    w.grad = torch.full((7, 3), float(rank + 1)); e.grad = torch.arange(60, dtype=torch.float32).reshape(4, 5, 3) * (rank + 1)
    # the reference's `light_base` is a plain leaf tensor with requires_grad (EnvironmentLight is not an nn.Module, nerf/utils.py:1853): rank 1 saw only
    # background and has no gradient for it — its VALUES must not enter the bucket; `raw` is an already-formed gradient tensor
    light = torch.full((3, 2, 3), 0.5 + rank, requires_grad=True)
    if rank == 0:
        light.grad = torch.full((3, 2, 3), 4.0)
    raw = torch.full((5,), 10.0 * (rank + 1))
    allreduce_gradients([w, e, unused, light, raw], algo=algo)
    # a bucket whose sum depends on the order of the additions (values 2^24 apart in magnitude) and whose length is no multiple of the slice size:
    # the direct exchange adds in rank order on whichever rank owns the slice, so every rank must hold the bits of ((g0 + g1) + g2)
    gen = torch.Generator().manual_seed(100 + rank)
    big = (torch.randn(1000, generator=gen) * torch.tensor([1.0, 3e7, 1e-7]).repeat(334)[:1000]).contiguous()
    mine = big.clone()
    allreduce_gradients([big], average=False, algo=algo)
    torch.save(dict(w=w.grad, e=e.grad, u=unused.grad, light=light.detach(), lg=light.grad, raw=raw, big=big, mine=mine), os.path.join(out, "g%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,world", [("direct", 3), ("direct", 2), ("allreduce", 2)])
def test_gradient_allreduce_gloo(tmp_path, algo, world):
    """Training exchange: parameter gradients averaged over ranks as one flat bucket; a parameter without a gradient on some rank counts as zero.
    'direct' (the default: all-to-all of slices, local sum in rank order, all-gather) on three ranks — an odd world, a bucket that needs padding — gives every
    rank the same bits, those of the rank-ordered sum; 'allreduce' is the single flat all-reduce of rounds 1-4."""
    import torch
    import torch.multiprocessing as mp
    port = 35500 + (os.getpid() % 2000) + (7 if algo == "direct" else 0) + world
    spawn_ranks(_grad_worker, (world, port, str(tmp_path), algo), world)
    ds = [torch.load(os.path.join(tmp_path, "g%d.pt" % r)) for r in range(world)]
    want = ds[0]["mine"].clone()
    for r in range(1, world):
        want = want + ds[r]["mine"]
    for r in range(world):
        assert torch.allclose(ds[r]["big"], want, rtol=1e-6, atol=0) and torch.equal(ds[r]["big"], ds[0]["big"])
        if algo == "direct":
            assert torch.equal(ds[r]["big"], want)
    if world != 2:
        return
    for r in range(2):
        d = ds[r]
        assert torch.allclose(d["w"], torch.full((7, 3), 1.5))
        assert torch.allclose(d["e"], torch.arange(60, dtype=torch.float32).reshape(4, 5, 3) * 1.5)
        assert torch.equal(d["u"], torch.zeros(2))
        assert torch.equal(d["light"], torch.full((3, 2, 3), 0.5 + r)) and torch.allclose(d["lg"], torch.full((3, 2, 3), 2.0))   # values untouched, missing grad = 0
        assert torch.allclose(d["raw"], torch.full((5,), 15.0))


def test_rgbe_roundtrip(tmp_path):
    from mirres_restir_nerf_mesh_amd import harness
    rng = np.random.default_rng(0)
    img = (rng.random((6, 10, 3)) * np.array([0.01, 1.0, 300.0])).astype(np.float32)
    img[0, 0] = 0
    p = str(tmp_path / "e.hdr")
    harness.write_hdr(p, img)
    back = harness.read_hdr(p)
    assert back.shape == img.shape and back.dtype == np.float32
    m = img.max(axis=2, keepdims=True)
    assert np.all(np.abs(back - img) <= m / 128.0 + 1e-9)      # 8-bit mantissa shared by the texel
    assert np.all(back[0, 0] == 0)


def test_bilateral_denoiser_rejects_a_wrong_channel_count():
    """The --use_bi_de filter reads colour 3 + normal 3 + (z, |dz|) per pixel; a [N, 7] input (a plain depth column instead of the pair,
    nerf/renderer.py:1080) would make the kernel read past the buffer, so the host wrapper refuses it before any launch."""
    import torch
    from mirres_restir_nerf_mesh_amd.renderutils.ops import bilateral_denoiser
    with pytest.raises(ValueError, match="8 channels"):
        bilateral_denoiser(4, 4, torch.zeros(16, 7), 2.0)


def test_camera_rays_and_shading_directions_against_the_reference():
    """harness.get_rays / view_dirs against nerf/utils.py:get_rays and render_stage1's `dirs` (scale_img_hwc(mag='nearest') + safe_normalize,
    nerf/renderer.py:935-946) run on the same pose (fixture ref_losses.npz)."""
    import torch
    from mirres_restir_nerf_mesh_amd import harness
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_losses.npz"))
    H, W = (int(v) for v in g["rays_hw"])
    ro, rd = harness.get_rays(torch.from_numpy(g["rays_pose"]), g["rays_intr"], H, W)
    np.testing.assert_allclose(ro.numpy(), g["rays_o"], rtol=0, atol=0)
    np.testing.assert_allclose(rd.numpy(), g["rays_d"], rtol=1e-6, atol=1e-7)
    d2 = harness.view_dirs(rd, H, W, 2)
    np.testing.assert_allclose(d2.numpy(), g["rays_dirs_ssaa2"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(harness.view_dirs(rd, H, W, 1).norm(dim=1).numpy(), 1.0, rtol=1e-6)


def test_antialias_topology_of_known_meshes():
    """raster.antialias_topology: for every edge (v_k, v_k+1) the vertex across it in the neighbouring triangle, -1 on a boundary (what dr.antialias
    needs to recognise silhouette edges; depends on the index buffer only)."""
    import torch
    from mirres_restir_nerf_mesh_amd.raster import antialias_topology
    one = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    assert antialias_topology(one).tolist() == [[-1, -1, -1]]
    quad = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)            # shared edge (0, 2): edge 2 of the first, edge 0 of the second
    assert antialias_topology(quad).tolist() == [[-1, -1, 3], [1, -1, -1]]
    tet = torch.tensor([[0, 1, 2], [0, 3, 1], [1, 3, 2], [2, 3, 0]], dtype=torch.int32)
    opp = antialias_topology(tet)
    assert opp.dtype == torch.int32 and (opp >= 0).all()
    for t in range(4):
        for k in range(3):
            a, b, c = int(tet[t, k]), int(tet[t, (k + 1) % 3]), int(tet[t, (k + 2) % 3])
            assert {a, b, c, int(opp[t, k])} == {0, 1, 2, 3}                  # closed tetrahedron: across every edge lies the fourth vertex
    from mirres_restir_nerf_mesh_amd import scene
    v, tri = scene.make_mesh(2, 2)                                           # closed sphere + an open ground grid (boundary edges)
    tri = torch.from_numpy(tri)
    opp = antialias_topology(tri).numpy()
    assert (opp < 0).any() and (opp >= 0).any()
    # every recorded opposite vertex belongs to another triangle that contains the edge
    t = tri.numpy()
    sets = [set(r) for r in t.tolist()]
    for i in range(0, len(t), max(1, len(t) // 40)):
        for k in range(3):
            o = int(opp[i, k])
            if o < 0:
                continue
            e = {int(t[i, k]), int(t[i, (k + 1) % 3])}
            assert any(j != i and e <= s and o in s for j, s in enumerate(sets))


def test_mvp_and_rays_describe_one_camera():
    """harness.mvp_from_pose (nerf/provider.py:277-288) against harness.get_rays (nerf/utils.py:350-423): a point on pixel (i, j)'s ray projects to
    NDC ((2i + 1) / W - 1, (2j + 1) / H - 1)."""
    import torch
    from mirres_restir_nerf_mesh_amd import harness
    H, W = 30, 50
    pose = torch.eye(4); pose[:3, :3] = torch.tensor([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]]); pose[:3, 3] = torch.tensor([0.3, -1.2, 2.0])
    intr = (61.0, 61.0, W * 0.5, H * 0.5)
    ro, rd = harness.get_rays(pose, intr, H, W)
    pts = ro + 2.7 * rd
    clip = torch.cat((pts, torch.ones(len(pts), 1)), 1) @ harness.mvp_from_pose(pose, intr, H, W).t()
    ndc = clip[:, :2] / clip[:, 3:4]
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    want = torch.stack(((2 * i + 1) / W - 1, (2 * j + 1) / H - 1), -1).view(-1, 2)
    assert float((ndc - want).abs().max()) < 1e-5 and bool((clip[:, 3] > 0).all())


def test_antialias_reference_statement_on_analytic_edges():
    """tests/util.py antialias_ref (the checker of the HIP operator) on cases with a closed-form answer: a half-plane-like triangle whose straight edge
    crosses pixel rows at a known abscissa — the blend weight of the partly covered pixel must equal its covered fraction along the row (and the
    pixel on the inside gives colour away when the edge cuts before the midpoint); a shared interior edge between two coplanar triangles blends nothing."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import antialias_ref
    H, W = 6, 16
    def clip_of(px, py):                    # pixel coordinates -> clip space with w = 1 (pixel centre (i, j) <-> NDC ((2i + 1) / W - 1, ...))
        return [2 * px / W - 1, 2 * py / H - 1, 0.0, 1.0]
    for edge_x in (10.3, 9.8):
        pos = np.array([clip_of(-50, -50), clip_of(edge_x, -50), clip_of(edge_x, 50), clip_of(-50, 50)], np.float64)
        tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
        opp = np.array([[-1, -1, 3], [1, -1, -1]], np.int32)        # edge (0, 2) is shared
        rast = np.zeros((H * W, 4)); color = np.zeros((H * W, 3))
        for p in range(H * W):
            x, y = (p % W) + 0.5, (p // W) + 0.5
            if x < edge_x:
                rast[p] = [0.3, 0.3, 1.0, 1 if (y - (-50)) * (edge_x + 50) < (x + 50) * 100 else 2]      # below / above the diagonal (0, 2)
                color[p] = [1.0, 0.5, 0.25]
        out, pairs = antialias_ref(color, rast, pos, tri, opp, H, W)
        inside = int(np.floor(edge_x - 0.5)); outside = inside + 1          # the pixel pair the edge passes between (centres at +0.5)
        u = edge_x - (inside + 0.5)
        for row in range(H):
            pi, po = row * W + inside, row * W + outside
            if u > 0.5:       # the triangle reaches into the outside pixel: it takes (u - 1/2) of the inside colour = its covered fraction of the row
                np.testing.assert_allclose(out[po], (u - 0.5) * color[pi], atol=1e-12); np.testing.assert_allclose(out[pi], color[pi], atol=1e-12)
            else:             # the edge cuts before the midpoint: the inside pixel is partly uncovered and gives (1/2 - u) away
                np.testing.assert_allclose(out[pi], color[pi] * (1 - (0.5 - u)), atol=1e-12); np.testing.assert_allclose(out[po], 0.0, atol=1e-12)
        touched = {p for p0, p1, a, _, _ in pairs for p in (p0, p1)}
        assert touched == {row * W + c for row in range(H) for c in (inside, outside)}          # only the silhouette pairs; the shared diagonal blends nothing
        changed = np.nonzero(np.abs(out - color).max(1) > 0)[0]
        assert set(changed.tolist()) <= touched


def test_the_two_software_rasterisers_of_the_gpu_tests_agree():
    """tests/util.py holds two float64 rasterisers the GPU tests compare raster.dr.rasterize with: one over projected 2-D triangles (no clipping; triangles with
    a vertex behind the eye skipped) and a homogeneous one (clipping by per-fragment z/w, analytic barycentric derivatives).  Where nothing needs clipping they
    must give the same record; the homogeneous one's derivatives must equal finite differences of its own barycentrics taken from a 9 x finer image (pixel
    (ix, iy) of the coarse image and pixel (9 ix + 4, 9 iy + 4) of the fine one share a centre; their neighbours lie 1/9 pixel away); and a triangle with one
    vertex behind the eye must show exactly its part beyond the near plane."""
    from util import rasterize_ref, rasterize_ref_homogeneous
    rng = np.random.default_rng(5)
    T = 30
    v = rng.uniform(-1, 1, (3 * T, 3)); t = np.arange(3 * T).reshape(T, 3)
    near, far = 0.5, 100.0
    proj = np.array([[1.2, 0, 0, 0], [0, -1.2, 0, 0], [0, 0, -(far + near) / (far - near), -2 * far * near / (far - near)], [0, 0, -1, 0]])
    view = np.eye(4); view[2, 3] = -3.0
    pc = np.concatenate([v, np.ones((3 * T, 1))], 1) @ (proj @ view).T
    H, W = 20, 24
    a, ea, ga = rasterize_ref(pc, t, H, W)
    b, db, eb, gb = rasterize_ref_homogeneous(pc, t, H, W)
    safe = ((ea > 1e-9) & (ga > 1e-9)) | (a[:, 3] == 0)
    assert (a[:, 3] > 0).mean() > 0.1 and safe.mean() > 0.95
    assert np.array_equal(a[safe, 3], b[safe, 3]) and np.abs(a[safe] - b[safe]).max() < 1e-12
    f, _, _, _ = rasterize_ref_homogeneous(pc, t, 9 * H, 9 * W)
    f = f.reshape(9 * H, 9 * W, 4); c = b.reshape(H, W, 4); d4 = db.reshape(H, W, 4)
    ctr = f[4::9, 4::9]; assert np.array_equal(ctr[..., 3], c[..., 3]) and np.abs(ctr - c).max() < 1e-12
    xm, xp, ym, yp = f[4::9, 3::9], f[4::9, 5::9], f[3::9, 4::9], f[5::9, 4::9]
    okx = (c[..., 3] > 0) & (xm[..., 3] == c[..., 3]) & (xp[..., 3] == c[..., 3]); oky = (c[..., 3] > 0) & (ym[..., 3] == c[..., 3]) & (yp[..., 3] == c[..., 3])
    fdx = (xp[..., :2] - xm[..., :2]) * 4.5; fdy = (yp[..., :2] - ym[..., :2]) * 4.5
    assert okx.sum() > 40 and oky.sum() > 40
    np.testing.assert_allclose(d4[okx][:, [0, 2]], fdx[okx], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(d4[oky][:, [1, 3]], fdy[oky], rtol=2e-3, atol=1e-6)
    # one big triangle through the eye plane: vertices at view depths 2, 2 and -1 (behind the eye)
    tri = np.array([[-1.0, -1.0, -2.0], [1.0, -1.0, -2.0], [0.0, 1.5, 1.0]])
    pc1 = np.concatenate([tri, np.ones((3, 1))], 1) @ proj.T
    r1, _, _, _ = rasterize_ref_homogeneous(pc1, np.array([[0, 1, 2]]), 40, 40)
    r0, _, _ = rasterize_ref(pc1, np.array([[0, 1, 2]]), 40, 40)
    assert (r0[:, 3] == 0).all() and (r1[:, 3] > 0).mean() > 0.2
    hit = r1[:, 3] > 0
    bw = np.stack([r1[:, 0], r1[:, 1], 1 - r1[:, 0] - r1[:, 1]], 1)
    depth = -(bw @ tri[:, 2])                                   # view depth of the point each covered pixel sees: beyond the near plane, never behind the eye
    assert (depth[hit] >= near - 1e-9).all() and (depth[hit] <= far).all() and depth[hit].min() < near + 0.05
    assert (np.abs(r1[hit, 2]) <= 1).all()
