"""GPU parity (through the C ABI): LBVH build and traversal are BIT-EXACT against the oracle."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


@pytest.mark.parametrize("subdiv,ground", [(0, 1), (2, 4), (4, 16), (5, 32)])
def test_build_bit_exact(torch_cuda, oracle, scene_mod, subdiv, ground):
    torch = torch_cuda
    from mirres_restir_nerf_mesh_amd.renderer_restir import restirbvhWorker
    v, t = scene_mod.make_mesh(subdiv, ground)
    w = restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda())
    w.update_mesh(w.vrt, w.v_ind)
    info, aabb, srt, _ = oracle.bvh_build(v, t)
    assert np.array_equal(w.LBVHNode_info.cpu().numpy(), info)
    assert np.array_equal(_bits(w.LBVHNode_aabb.cpu().numpy()), _bits(aabb))
    # structural invariants, independent of the oracle: every internal box is the union of its children, each leaf reachable once
    a = w.LBVHNode_aabb.cpu().numpy(); i = w.LBVHNode_info.cpu().numpy(); T = len(t)
    L, R = i[:T - 1, 0], i[:T - 1, 1]
    assert np.array_equal(a[:T - 1, :3], np.minimum(a[L, :3], a[R, :3])) and np.array_equal(a[:T - 1, 3:], np.maximum(a[L, 3:], a[R, 3:]))
    assert sorted(np.concatenate([L, R]).tolist()) == list(range(1, 2 * T - 1))
    assert sorted(i[T - 1:, 2].tolist()) == list(range(T))


@pytest.mark.parametrize("T", [2, 3, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4103, 10000, 70001])
def test_radix_sort_is_stable_across_wave_and_tile_boundaries(torch_cuda, oracle, T):
    """a-3 (lbvh_single_radixsort.slang): ascending Morton codes, equal codes in ascending element order. The hand-written sort works on 2048-key tiles of four
    512-key wave chunks taken 64 keys at a time; sizes around those boundaries, with few distinct cells so that every tile holds long runs of equal codes.
    Checked through the C ABI's `sorted_codes` output against the oracle's build and against numpy's stable sort of the codes themselves."""
    torch = torch_cuda
    import ctypes as C
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    rng = np.random.default_rng(T)
    cells = rng.random((max(2, T // 40), 3)).astype(np.float32) * 1.8 - 0.9          # ~40 triangles per Morton cell
    c = cells[rng.integers(0, len(cells), T)]
    tri0 = np.array([[0, 0, 0], [3e-5, 1e-5, 2e-5], [1e-5, 3e-5, 2e-5]], np.float32)
    v = (c[:, None, :] + tri0[None] + rng.random((T, 1, 3)).astype(np.float32) * 1e-6).reshape(-1, 3).astype(np.float32)
    v[0] = (-1, -1, -1); v[-1] = (1, 1, 1)                                            # the scene extent
    t = np.arange(3 * T, dtype=np.int32).reshape(-1, 3)
    h = C.c_void_p()
    check(lib().mirres_bvh_create(C.byref(h), T), "create")
    try:
        dv, dt = torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()
        srt = torch.full((T, 2), -1, dtype=torch.int32, device="cuda")
        for _ in range(2):                                                            # a second build reuses the ping-pong buffers and counters
            check(lib().mirres_bvh_build(h, dv.data_ptr(), 3 * T, dt.data_ptr(), T, None, None, srt.data_ptr(), None), "build")
        torch.cuda.synchronize()
        got = srt.cpu().numpy()
    finally:
        lib().mirres_bvh_destroy(h)
    _, _, ref, _ = oracle.bvh_build(v, t)
    assert np.array_equal(got, ref)
    codes = np.empty(T, np.int64); codes[ref[:, 1]] = ref[:, 0]                       # code of every element
    order = np.argsort(codes, kind="stable")
    assert np.array_equal(got[:, 1], order) and np.array_equal(got[:, 0], codes[order])
    assert T < 100 or len(np.unique(codes)) < T // 8                                  # long runs of equal codes


def test_duplicate_morton_codes(torch_cuda, oracle):
    """Many triangles in one Morton cell: ties are broken by sorted position (lbvh_hierarchy.slang:47-48), so the sort must be stable."""
    torch = torch_cuda
    from mirres_restir_nerf_mesh_amd.renderer_restir import restirbvhWorker
    rng = np.random.default_rng(3)
    base = np.array([[0, 0, 0], [1e-4, 2e-5, 3e-5], [2e-5, 1e-4, 5e-5]], np.float32)
    n = 300
    v = np.concatenate([base + rng.random((1, 3)).astype(np.float32) * 1e-5 for _ in range(n)] + [np.array([[-1, -1, -1], [1, 1.1, 1.2], [0.9, -1, 0.3]], np.float32)])
    t = np.arange(3 * (n + 1), dtype=np.int32).reshape(-1, 3)
    w = restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); w.update_mesh(w.vrt, w.v_ind)
    info, aabb, _, _ = oracle.bvh_build(v, t)
    assert np.array_equal(w.LBVHNode_info.cpu().numpy(), info) and np.array_equal(_bits(w.LBVHNode_aabb.cpu().numpy()), _bits(aabb))


@pytest.mark.parametrize("subdiv,ground,hw", [(3, 16, 96), (5, 32, 160)])
def test_trace_bit_exact(torch_cuda, oracle, scene_mod, subdiv, ground, hw):
    torch = torch_cuda
    from mirres_restir_nerf_mesh_amd.renderer_restir import restirbvhWorker
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    v, t = scene_mod.make_mesh(subdiv, ground)
    w = restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); w.update_mesh(w.vrt, w.v_ind)
    info, aabb, _, _ = oracle.bvh_build(v, t)
    eye, rd = scene_mod.camera_rays(hw, hw)
    n = hw * hw
    prim = oracle.trace(info, aabb, v, t, oracle.make_rays(np.repeat(eye[None], n, 0), rd), True)
    rng = np.random.default_rng(7)
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    d2[::17, 0] = 0.0  # exercises the zero-component fix-up (helperDi.slang:153-154)
    o2 = (prim["pos"] + 0.01 * d2).astype(np.float32)
    sets = {"primary": oracle.make_rays(np.repeat(eye[None], n, 0), rd), "secondary": oracle.make_rays(o2[prim["hit"] > 0], d2[prim["hit"] > 0]),
            "short": oracle.make_rays(o2[prim["hit"] > 0], d2[prim["hit"] > 0], 0.0, 0.3)}
    for name, rays in sets.items():
        k = len(rays)
        ref = oracle.trace(info, aabb, v, t, rays, True, True)
        assert ref["counters"][:, 3].sum() == 0, "fixture must not overflow the 64-entry stack"
        dr = torch.from_numpy(rays).cuda()
        hit = torch.zeros(k, dtype=torch.int32, device="cuda"); tt = torch.zeros(k, device="cuda"); pos = torch.zeros((k, 3), device="cuda")
        nrm = torch.zeros((k, 3), device="cuda"); pr = torch.zeros(k, dtype=torch.int32, device="cuda"); cnt = torch.zeros((k, 4), dtype=torch.int32, device="cuda")
        check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), k, 1, hit.data_ptr(), tt.data_ptr(), pos.data_ptr(), nrm.data_ptr(), pr.data_ptr(), cnt.data_ptr(), None), name)
        torch.cuda.synchronize()
        assert np.array_equal(hit.cpu().numpy(), ref["hit"]), name
        assert np.array_equal(pr.cpu().numpy(), ref["prim"]), name              # BVH hit indices bit-exact (north star)
        assert np.array_equal(_bits(tt.cpu().numpy()), _bits(ref["t"])), name
        assert np.array_equal(_bits(pos.cpu().numpy()), _bits(ref["pos"])), name
        assert np.array_equal(_bits(nrm.cpu().numpy()), _bits(ref["normal"])), name
        assert np.array_equal(cnt.cpu().numpy()[:, :3].astype(np.uint32), ref["counters"][:, :3]), name   # same nodes visited
        # mode 2: front-to-back 4-wide fast path + reference-order redo of order-dependent rays -> still bit-exact
        h2 = torch.zeros(k, dtype=torch.int32, device="cuda"); t2 = torch.zeros(k, device="cuda"); p2 = torch.zeros((k, 3), device="cuda")
        n2 = torch.zeros((k, 3), device="cuda"); pr2 = torch.zeros(k, dtype=torch.int32, device="cuda")
        check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), k, 2, h2.data_ptr(), t2.data_ptr(), p2.data_ptr(), n2.data_ptr(), pr2.data_ptr(), None, None), name)
        torch.cuda.synchronize()
        assert np.array_equal(h2.cpu().numpy(), ref["hit"]), name + " fast"
        assert np.array_equal(pr2.cpu().numpy(), ref["prim"]), name + " fast"
        assert np.array_equal(_bits(t2.cpu().numpy()), _bits(ref["t"])), name + " fast"
        assert np.array_equal(_bits(p2.cpu().numpy()), _bits(ref["pos"])), name + " fast"
        assert np.array_equal(_bits(n2.cpu().numpy()), _bits(ref["normal"])), name + " fast"
        hit0 = torch.zeros(k, dtype=torch.int32, device="cuda")
        check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), k, 0, hit0.data_ptr(), None, None, None, None, None, None), name)
        assert np.array_equal(hit0.cpu().numpy(), ref["hit"]), name + " any-hit"
        # independent invariant: a reported hit is a triangle brute-force Moller-Trumbore also accepts (BVH subset of brute force)
    # empty batch and bad mode
    assert lib().mirres_bvh_trace(w.h, dr.data_ptr(), 0, 1, None, None, None, None, None, None, None) == 0
    assert lib().mirres_bvh_trace(w.h, dr.data_ptr(), 4, 7, None, None, None, None, None, None, None) < 0


def test_traversal_quirks_on_a_hand_made_mesh(torch_cuda, oracle):
    """The edge cases of SURVEY §8 a-7..a-9 on a mesh built for them, every trace mode against the oracle bit for bit: axis-aligned flat triangles (their
    boxes have zero thickness and can never be entered: `tmax <= tmin`), duplicated triangles (exact distance ties: the ordered fast path must hand them
    to the reference-order redo), a zero-area triangle (|det| < 1e-15), hits behind the origin (triangle_hit does not look at t), rays with zero
    direction components and axis-parallel rays, origins on vertices / edges, tiny and huge t_max."""
    torch = torch_cuda
    from mirres_restir_nerf_mesh_amd.renderer_restir import restirbvhWorker
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    rng = np.random.default_rng(12)
    V, T = [], []
    def quad(a, b, c, d):
        i = len(V); V.extend([a, b, c, d]); T.extend([[i, i + 1, i + 2], [i, i + 2, i + 3]])
    # a unit cube of axis-aligned faces (never hit by the reference), two tilted walls, a tilted floor, duplicates of one wall, a degenerate sliver
    for z in (0.0, 1.0): quad([0, 0, z], [1, 0, z], [1, 1, z], [0, 1, z])
    for x in (0.0, 1.0): quad([x, 0, 0], [x, 1, 0], [x, 1, 1], [x, 0, 1])
    for y in (0.0, 1.0): quad([0, y, 0], [1, y, 0], [1, y, 1], [0, y, 1])
    quad([-1, -1, 0.2], [2, -1, 0.25], [2.1, 2, 0.3], [-1.05, 2, 0.22])
    wall = ([0.3, -0.5, -0.5], [0.35, 1.5, -0.5], [0.4, 1.5, 1.5], [0.33, -0.5, 1.5])
    quad(*wall); quad(*wall); quad(*wall)                                   # three coincident walls: exact ties
    quad([0.7, -0.5, -0.4], [0.72, 1.4, -0.5], [0.9, 1.5, 1.6], [0.75, -0.6, 1.5])
    i = len(V); V.extend([[0.5, 0.5, 0.5], [0.5, 0.5, 0.5], [0.6, 0.6, 0.6]]); T.append([i, i + 1, i + 2])   # zero area
    for _ in range(40):                                                     # filler so that the hierarchy has some depth
        c = rng.random(3) * 3 - 1; e = (rng.random((3, 3)) - 0.5) * 0.4
        i = len(V); V.extend([c, c + e[0], c + e[1]]); T.append([i, i + 1, i + 2])
    v = np.array(V, np.float32); t = np.array(T, np.int32)
    w = restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); w.update_mesh(w.vrt, w.v_ind)
    info, aabb, _, _ = oracle.bvh_build(v, t)
    assert np.array_equal(w.LBVHNode_info.cpu().numpy(), info) and np.array_equal(w.LBVHNode_aabb.cpu().numpy(), aabb)
    n = 6000
    o = (rng.random((n, 3)) * 3 - 1).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32)
    d[0::7, 0] = 0; d[1::7, 1] = 0; d[2::7, 2] = 0; d[3::49] = [0, 0, 1]; d[4::49] = [1, 0, 0]; d[5::49] = [0, -1, 0]   # zero components, axis-parallel
    o[6::11] = v[rng.integers(0, len(v), size=len(o[6::11]))]                                                          # origins on vertices
    o[7::13] = (0.5 * (v[t[rng.integers(0, len(t), size=len(o[7::13])), 0]] + v[t[rng.integers(0, len(t), size=len(o[7::13])), 1]])).astype(np.float32)
    o[8::5, 0] = 0.5; o[8::5, 1] = 0.5; o[8::5, 2] = 0.5                                                               # inside the cube, between the walls
    tmax = np.full(n, 1e7, np.float32); tmax[::9] = 0.05; tmax[1::9] = 1e-4
    rays = oracle.make_rays(o, d); rays[:, 7] = tmax
    ref = oracle.trace(info, aabb, v, t, rays, True, True)
    assert (ref["hit"] > 0).mean() > 0.3 and (ref["t"][ref["hit"] > 0] < 0).any()         # behind-the-origin hits occur
    dr = torch.from_numpy(rays).cuda()
    for mode in (1, 2):
        hit = torch.zeros(n, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, device="cuda"); pos = torch.zeros((n, 3), device="cuda")
        nrm = torch.zeros((n, 3), device="cuda"); pr = torch.zeros(n, dtype=torch.int32, device="cuda")
        check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, mode, hit.data_ptr(), tt.data_ptr(), pos.data_ptr(), nrm.data_ptr(), pr.data_ptr(), None, None), "mode %d" % mode)
        torch.cuda.synchronize()
        m = ref["hit"] > 0
        assert np.array_equal(hit.cpu().numpy(), ref["hit"]) and np.array_equal(pr.cpu().numpy(), ref["prim"]), mode
        assert np.array_equal(_bits(tt.cpu().numpy()[m]), _bits(ref["t"][m])) and np.array_equal(_bits(nrm.cpu().numpy()[m]), _bits(ref["normal"][m])), mode
    h0 = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, 0, h0.data_ptr(), None, None, None, None, None, None), "any")
    assert np.array_equal(h0.cpu().numpy(), ref["hit"])
    h3 = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, 3, h3.data_ptr(), None, None, None, None, None, None), "front")
    assert np.array_equal(h3.cpu().numpy(), oracle.occluded_front(info, aabb, v, t, rays))
    # the axis-aligned cube faces (triangles 0..11) are never reported: their boxes cannot be entered
    assert not np.isin(ref["prim"][ref["hit"] > 0], np.arange(12)).any()
    # mode 4, the conventional closest hit (nearest triangle at 0 < t <= t_max; mirres_rasterize's near-plane rays): same bit as the front-only occlusion
    # query on the long rays, and the reported triangle is the nearest one a float64 Moeller-Trumbore over ALL triangles (the un-enterable cube faces
    # aside) finds in front of the origin
    hit = torch.zeros(n, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, device="cuda"); pr = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, 4, hit.data_ptr(), tt.data_ptr(), None, None, pr.data_ptr(), None, None), "mode 4")
    h4, t4, p4 = hit.cpu().numpy() > 0, tt.cpu().numpy().astype(np.float64), pr.cpu().numpy()
    long_ray = tmax > 1
    assert np.array_equal(h4[long_ray], h3.cpu().numpy()[long_ray] > 0)
    od = o.astype(np.float64); dd = d.astype(np.float64); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    tb = np.full((n, len(t)), np.inf)
    for k in range(12, len(t)):
        a, e1, e2 = (v[t[k, 0]].astype(np.float64), v[t[k, 1]].astype(np.float64) - v[t[k, 0]], v[t[k, 2]].astype(np.float64) - v[t[k, 0]])
        P = np.cross(dd, e2); det = P @ e1
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1 / det; Tv = od - a; uu = (Tv * P).sum(1) * inv; Q = np.cross(Tv, e1); vv = (dd * Q).sum(1) * inv; tk = (Q @ e2) * inv
        eps = 1e-5
        ok = (np.abs(det) > 1e-12) & (uu > eps) & (vv > eps) & (uu + vv < 1 - eps) & (tk > 1e-4)        # clearly inside, clearly in front
        tb[ok, k] = tk[ok]
    nearest = tb.min(1)
    sure = np.isfinite(nearest) & (nearest < tmax * 0.99)
    assert h4[sure].all() and sure.sum() > 1000
    assert (t4[h4] > 0).all() and (t4[h4] <= tmax[h4]).all() and (p4[h4] >= 12).all()
    assert (t4[sure] <= nearest[sure] * (1 + 1e-4) + 1e-5).all()                                     # nothing clearly in front of the reported hit
    rows = np.nonzero(sure)[0]
    own = tb[rows, p4[rows]]                                                                          # the reported triangle's own float64 distance
    agree = np.isfinite(own)
    assert agree.mean() > 0.95 and np.allclose(own[agree], t4[rows][agree], rtol=1e-4, atol=1e-5)
    assert (~h4[(~np.isfinite(tb).any(1)) & ~(oracle.occluded_front(info, aabb, v, t, rays) > 0)]).all()


def test_fused_interior_slab_test_on_hostile_geometry(torch_cuda, oracle):
    """The compressed 4-wide tree's interior boxes are tested with one fma per plane, made conservative by a per-ray margin (bvh_trace.hip, node_cons):
    every trace mode must still equal the oracle bit for bit where that margin matters — clusters of near-coincident tiny triangles (nodes far thinner
    than 2^-19 of the scene: unused child slots could pass the fused test), a scene spanning +-100 with origins up to 1e3 away, directions with
    denormal or zero components (infinite reciprocals) and grazing rays along the cluster planes."""
    torch = torch_cuda
    from mirres_restir_nerf_mesh_amd.renderer_restir import restirbvhWorker
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    rng = np.random.default_rng(21)
    V, T = [], []
    centres = (rng.random((24, 3)) * 200 - 100).astype(np.float64)
    for c in centres:
        k = int(rng.integers(1, 7))                                   # 1..6 triangles per cluster: nodes with 1, 2, 3 real children occur
        for _ in range(k):
            e = rng.normal(size=(3, 3)) * rng.choice([1e-7, 1e-5, 1e-3])
            i = len(V); V.extend([c + e[0], c + e[1], c + e[2]]); T.append([i, i + 1, i + 2])
    for _ in range(300):                                              # ordinary geometry around them
        c = rng.random(3) * 200 - 100; e = rng.normal(size=(3, 3)) * 6.0
        i = len(V); V.extend([c, c + e[0], c + e[1]]); T.append([i, i + 1, i + 2])
    v = np.array(V, np.float32); t = np.array(T, np.int32)
    w = restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); w.update_mesh(w.vrt, w.v_ind)
    info, aabb, _, _ = oracle.bvh_build(v, t)
    assert np.array_equal(w.LBVHNode_info.cpu().numpy(), info) and np.array_equal(w.LBVHNode_aabb.cpu().numpy(), aabb)
    n = 20000
    o = (rng.random((n, 3)) * 240 - 120).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32)
    tgt = v[t[rng.integers(0, len(t), size=n), 0]]                    # most rays aim at (or start from) a triangle: hits are common
    aim = rng.random(n) < 0.7
    d[aim] = (tgt[aim] + rng.normal(size=(int(aim.sum()), 3)).astype(np.float32) * 1e-3) - o[aim]
    o[0::17] = tgt[0::17] + (rng.normal(size=(len(o[0::17]), 3)) * 1e-6).astype(np.float32)     # origins inside the tiny clusters
    o[1::19] *= 8.0                                                    # far outside the scene box
    d[2::23, 0] = np.float32(1e-40); d[3::23, 1] = np.float32(-1e-42); d[4::23, 2] = 0.0         # denormal / zero components
    d[5::29] = [1e-39, 1.0, 0.0]
    rays = oracle.make_rays(o, d)
    ref = oracle.trace(info, aabb, v, t, rays, True, True)
    assert 0.1 < (ref["hit"] > 0).mean() < 0.95
    dr = torch.from_numpy(rays).cuda()
    for mode in (1, 2):
        hit = torch.zeros(n, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, device="cuda"); pos = torch.zeros((n, 3), device="cuda")
        nrm = torch.zeros((n, 3), device="cuda"); pr = torch.zeros(n, dtype=torch.int32, device="cuda")
        check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, mode, hit.data_ptr(), tt.data_ptr(), pos.data_ptr(), nrm.data_ptr(), pr.data_ptr(), None, None), "mode %d" % mode)
        torch.cuda.synchronize()
        m = ref["hit"] > 0
        assert np.array_equal(hit.cpu().numpy(), ref["hit"]) and np.array_equal(pr.cpu().numpy(), ref["prim"]), mode
        assert np.array_equal(_bits(tt.cpu().numpy()[m]), _bits(ref["t"][m])), mode
    h0 = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, 0, h0.data_ptr(), None, None, None, None, None, None), "any")
    assert np.array_equal(h0.cpu().numpy(), ref["hit"])
    h3 = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, 3, h3.data_ptr(), None, None, None, None, None, None), "front")
    assert np.array_equal(h3.cpu().numpy(), oracle.occluded_front(info, aabb, v, t, rays))


def test_worker_is_released_when_dropped(torch_cuda, scene_mod):
    """The registry that maps LBVHNode_info back to its worker (the reference's launch wrappers pass the node arrays, not the worker) holds weak
    references: a worker the caller drops is destroyed — its native BVH freed — instead of living until the process ends."""
    import gc, weakref
    torch = torch_cuda
    from mirres_restir_nerf_mesh_amd import Resampling
    from mirres_restir_nerf_mesh_amd.renderer_restir import restirbvhWorker
    v, t = scene_mod.make_mesh(2, 2)
    n0 = len(Resampling._BVH_OWNERS)
    w = restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); w.update_mesh(w.vrt, w.v_ind)
    info = w.LBVHNode_info
    assert len(Resampling._BVH_OWNERS) == n0 + 1 and Resampling._owner(info) is w
    w.update_mesh(w.vrt, w.v_ind)                      # a rebuild re-registers under the (possibly new) array, never twice
    assert len(Resampling._BVH_OWNERS) == n0 + 1
    info = w.LBVHNode_info
    r = weakref.ref(w); del w; gc.collect()
    assert r() is None and len(Resampling._BVH_OWNERS) == n0
    from mirres_restir_nerf_mesh_amd._lib import MirresError
    with pytest.raises(MirresError, match="live restirbvhWorker"):
        Resampling._owner(info)


def test_deep_tree_private_stack_and_reference_stack(torch_cuda, oracle):
    """The adversarial mesh of tests/test_oracle_invariants.py (a 27-deep right-running Morton chain + 65 536 triangles of ONE Morton code, every leaf box around
    one common line, no triangle on it): the LBVH is ~43 deep, the reference's stack holds depth + 1 entries (< 64: it cannot overflow for any int32 T), and the
    4-wide private stack of the shadow-ray kernel — up to three deferred references per 4-wide level — goes DEEPER than the reference's 64 would allow: the
    kernel's stack is 256 entries >= 3 x (40 + 14 + 31) (bvh_trace.hip MR_ANY_STACK: SAH top + extended-Morton clusters), so nothing can be dropped.  Every mode of mirres_bvh_trace equals
    the oracle on the line itself and on rays jittered around it (hits and misses)."""
    torch = torch_cuda
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_invariants import adversarial_chain_mesh
    from mirres_restir_nerf_mesh_amd.renderer_restir import restirbvhWorker
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    v, t, (lx, ly) = adversarial_chain_mesh(65536)
    w = restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); w.update_mesh(w.vrt, w.v_ind)
    info, aabb, _, _ = oracle.bvh_build(v, t)
    assert np.array_equal(w.LBVHNode_info.cpu().numpy(), info) and np.array_equal(_bits(w.LBVHNode_aabb.cpu().numpy()), _bits(aabb))
    rng = np.random.default_rng(5)
    n = 4096
    o = np.tile(np.array([[1.0 - lx, 1.0 - ly, -1.5]], np.float32), (n, 1)); d = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (n, 1))
    o[1:, :2] += (rng.random((n - 1, 2)).astype(np.float32) - 0.5) * np.float32(0.004)        # around the line: inside and outside the triangles' half
    d[n // 2:, :2] += (rng.random((n - n // 2, 2)).astype(np.float32) - 0.5) * np.float32(0.002)
    rays = oracle.make_rays(o, d)
    ref = oracle.trace(info, aabb, v, t, rays, True, counters=True)
    assert ref["hit"][0] == 0 and ref["counters"][0, 2] > 60000 and ref["counters"][:, 3].sum() == 0 and 0.05 < ref["hit"].mean() < 0.95
    depth = oracle.tree_depth(info); deepest_ref = int(oracle.trace_stack_depth(info, aabb, v, t, rays).max())
    assert 40 <= depth <= 30 + int(np.ceil(np.log2(len(t)))) and depth <= deepest_ref <= depth + 1 < 64
    dr = torch.from_numpy(rays).cuda()
    for mode in (0, 1, 2):
        hit = torch.zeros(n, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, device="cuda"); p = torch.zeros((n, 3), device="cuda")
        nn = torch.zeros((n, 3), device="cuda"); pr = torch.zeros(n, dtype=torch.int32, device="cuda")
        check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, mode, hit.data_ptr(), tt.data_ptr(), p.data_ptr(), nn.data_ptr(), pr.data_ptr(), None, None), "trace")
        torch.cuda.synchronize()
        assert np.array_equal(hit.cpu().numpy(), ref["hit"]), mode
        if mode:
            m = ref["hit"] > 0
            assert np.array_equal(pr.cpu().numpy(), ref["prim"]) and np.array_equal(_bits(tt.cpu().numpy()[m]), _bits(ref["t"][m])) and np.array_equal(_bits(nn.cpu().numpy()[m]), _bits(ref["normal"][m]))
    L = lib(); L.mirres_debug_any_stats.argtypes = [C.c_void_p] * 6; L.mirres_debug_any_stats.restype = C.c_int
    st = (C.c_uint64 * 12)(); hit = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(L.mirres_debug_any_stats(w.h, dr.data_ptr(), n, hit.data_ptr(), st, None), "any stats")
    assert np.array_equal(hit.cpu().numpy(), ref["hit"])
    bound = 3 * (54 + int(np.ceil(np.log2(len(t)))))
    assert st[11] == 0 and 20 <= st[8] <= bound <= 256, (st[8], bound)
    rep = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(rep):
        open(os.path.join(rep, "deep_tree_stack.txt"), "w").write(
            "adversarial chain mesh T=%d: LBVH depth %d; reference stack deepest %d of 64; shadow-ray kernel's private stack deepest %d (bound 3 x (54 + ceil(log2 T)) = %d, capacity 256); overflows %d; "
            "64-byte records per ray %.1f\n" % (len(t), depth, deepest_ref, st[8], bound, st[11], st[3] / n))
