"""CPU: invariants that pin the oracle independently of its own code path: brute-force intersection, LBVH structure,
Monte-Carlo convergence of the ReSTIR-DI estimator to a direct quadrature of the rendering integral, hash-grid vs dense trilinear."""
import numpy as np
import pytest

from util import SmallFrame


def _brute(v, t, o, d):
    """Moller-Trumbore over every triangle with the reference's acceptance rule (no t interval), float64."""
    v0 = v[t[:, 0]].astype(np.float64); e1 = v[t[:, 1]].astype(np.float64) - v0; e2 = v[t[:, 2]].astype(np.float64) - v0
    out = []
    for oi, di in zip(o.astype(np.float64), d.astype(np.float64)):
        P = np.cross(di, e2); det = (e1 * P).sum(1); ok = np.abs(det) >= 1e-15
        inv = 1 / np.where(ok, det, 1); T = oi - v0; u = (T * P).sum(1) * inv; Q = np.cross(T, e1); w = (di * Q).sum(1) * inv; tt = (e2 * Q).sum(1) * inv
        m = ok & (u >= 0) & (u <= 1) & (w >= 0) & (u + w <= 1)
        out.append((m, tt))
    return out


def test_bvh_structure_and_bruteforce(oracle, scene_mod):
    v, t = scene_mod.make_mesh(3, 8)
    info, aabb, srt, h = oracle.bvh_build(v, t)
    T = len(t)
    assert np.all(np.diff(srt[:, 0].astype(np.int64)) >= 0) and sorted(srt[:, 1].tolist()) == list(range(T))
    eq = srt[1:, 0] == srt[:-1, 0]
    assert np.all(srt[1:, 1][eq] > srt[:-1, 1][eq])                       # stable: ties keep element order
    L, R = info[:T - 1, 0], info[:T - 1, 1]
    assert np.array_equal(aabb[:T - 1, :3], np.minimum(aabb[L, :3], aabb[R, :3])) and np.array_equal(aabb[:T - 1, 3:], np.maximum(aabb[L, 3:], aabb[R, 3:]))
    assert sorted(np.concatenate([L, R]).tolist()) == list(range(1, 2 * T - 1))
    tv = v[t[info[T - 1:, 2]]]
    assert np.array_equal(aabb[T - 1:, :3], tv.min(1)) and np.array_equal(aabb[T - 1:, 3:], tv.max(1))
    eye, rd = scene_mod.camera_rays(24, 24)
    n = 576
    o = np.repeat(eye[None], n, 0); d = rd / np.linalg.norm(rd, axis=1, keepdims=True)
    r = oracle.trace(info, aabb, v, t, oracle.make_rays(o, rd), True, True)
    br = _brute(v, t, o, d.astype(np.float32))
    for i, (m, tt) in enumerate(br):
        if r["hit"][i]:
            assert m[r["prim"][i]], "BVH hit must be a triangle brute force accepts"       # BVH subset of brute force
        if m.any() and tt[m].min() > 1e-3:                                                  # all line hits in front: closest t equals the brute-force minimum
            assert r["hit"][i] and abs(r["t"][i] - tt[m].min()) < 1e-4
        if not m.any():
            assert not r["hit"][i]
    assert r["counters"][:, 3].sum() == 0
    anyr = oracle.trace(info, aabb, v, t, oracle.make_rays(o, rd), False)
    assert np.array_equal(anyr["hit"], r["hit"])


def test_clustered_mesh_is_what_it_claims(scene_mod):
    """scene.make_mesh_clustered (the lego-like workload): ~3.3e5 triangles inside the unit bound, seeded, triangle areas spread far beyond 100 : 1,
    and a few hundred exactly axis-aligned triangles (zero-thickness leaf boxes, helperDi.slang:165)."""
    v, t = scene_mod.mesh_by_name("clustered")
    assert 300000 <= len(t) <= 345000 and np.abs(v).max() < 1.0
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross((b - a).astype(np.float64), (c - a).astype(np.float64)), axis=1)
    assert (area > 0).all() and area.max() / area.min() > 1e4
    flat = (np.maximum(np.maximum(a, b), c) - np.minimum(np.minimum(a, b), c)).min(axis=1) == 0
    assert 100 < int(flat.sum()) < 2000
    v2, t2 = scene_mod.mesh_by_name("clustered")
    assert np.array_equal(v, v2) and np.array_equal(t, t2)
    small_v, small_t = scene_mod.make_mesh_clustered(13000)
    assert 10000 <= len(small_t) <= 20000          # the floor: stud and part counts do not scale, only tessellations do


def test_clustered_mesh_bvh_against_brute_force_and_the_flat_box_quirk(oracle, scene_mod):
    """On the small clustered mesh: every BVH hit is a triangle brute force accepts; where brute force accepts nothing the BVH misses; and a triangle whose
    leaf box has zero thickness (the axis-aligned plates) is NEVER reported although brute force accepts it — aabb_hit rejects t_max <= t_min (helperDi.slang:165)."""
    v, t = scene_mod.make_mesh_clustered(13000)
    info, aabb, srt, h = oracle.bvh_build(v, t)
    T = len(t)
    L, R = info[:T - 1, 0], info[:T - 1, 1]
    assert np.array_equal(aabb[:T - 1, :3], np.minimum(aabb[L, :3], aabb[R, :3])) and np.array_equal(aabb[:T - 1, 3:], np.maximum(aabb[L, 3:], aabb[R, 3:]))
    tv = v[t]
    flat = (tv.max(1) - tv.min(1)).min(axis=1) == 0
    assert flat.sum() > 100
    eye, rd = scene_mod.camera_rays(28, 28)
    n = 28 * 28
    o = np.repeat(eye[None], n, 0)
    # a second bundle straight down onto the axis-aligned plates
    ctr = tv[flat].mean(1)[::max(1, int(flat.sum()) // 60)]
    o2 = ctr + np.array([0.003, 0.002, 0.5], np.float32); d2 = np.repeat(np.array([[0.0, 0.0, -1.0]], np.float32), len(o2), 0)
    o = np.concatenate([o, o2]).astype(np.float32); dd = np.concatenate([rd, d2]).astype(np.float32)
    r = oracle.trace(info, aabb, v, t, oracle.make_rays(o, dd), True, True)
    dn = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    br = _brute(v, t, o, dn.astype(np.float32))
    saw_flat_only = 0
    for i, (m, tt) in enumerate(br):
        if r["hit"][i]:
            assert m[r["prim"][i]] and not flat[r["prim"][i]]
        if not m.any():
            assert not r["hit"][i]
        if m.any() and not (m & ~flat).any():
            saw_flat_only += 1
            assert not r["hit"][i], "only zero-thickness-box triangles on this line: the reference can never enter their leaves"
    assert saw_flat_only > 0 and r["counters"][:, 3].sum() == 0


def adversarial_chain_mesh(dups=64):
    """Deep LBVH + a line through every leaf box that misses every triangle.  The scene extent is [-1, 1]^3; triangle k (k = 1..27) has its box centre in the
    cell whose Morton code has the top bit of every axis and the next k bits set (x, y, z cells 1024 - 2^(9-j)): sorted, every split peels ONE leaf off to the
    left and the chain continues to the right — bvh_hit pushes left then right (helperDi.slang:245-246), so the lefts pile up.  `dups` more triangles share
    the deepest cell (identical codes: told apart by sorted position only).  Every box is [2c - 1, 1] per axis, hence contains the line (1 - lx, 1 - ly, .);
    each triangle covers the half of its x / y rectangle the line does not pass.  Returns (verts, tris, (lx, ly))."""
    tris = []
    lx, ly = 0.0004, 0.0014
    def add(cell_x, cell_y, cell_z, jitter=0.0):
        c = [2.0 * (q + 0.5) / 1024.0 - 1.0 for q in (cell_x, cell_y, cell_z)]
        x0, y0, z0 = 2 * c[0] - 1, 2 * c[1] - 1, 2 * c[2] - 1
        tris.append([[1.0, 1.0, 1.0 + jitter], [x0, y0, z0], [x0, 1.0, 0.5 * (1 + z0)]])
    hb = lambda n: 1024 - (1 << (9 - n))
    for k in range(1, 28):       # bit 3i+2 of the code is x's bit i, 3i+1 y's, 3i z's: the next k bits set = x, then y, then z cells
        j, r_ = divmod(k, 3)
        add(hb(j + (1 if r_ >= 1 else 0)), hb(j + (1 if r_ >= 2 else 0)), hb(j))
    for q in range(dups):
        add(1023, 1023, 1023, jitter=-1e-7 * q)
    tris.append([[1, 1, 1], [-1, -1, -1], [-1, 1, 0]])                # fixes the scene extent at [-1, 1]^3
    v = np.asarray(tris, np.float32).reshape(-1, 3); t = np.arange(len(v), dtype=np.int32).reshape(-1, 3)
    return v, t, (lx, ly)


def test_reference_stack_cannot_overflow(oracle):
    """The 64-entry stack of bvh_hit (helperDi.slang:136) holds at most depth + 1 entries, and the LBVH's depth is bounded by the bits of its sort key:
    30 Morton bits + ceil(log2 T) position bits (lbvh_hierarchy.slang:40-60) <= 61 for any int32 T.  An adversarial mesh — a 30-deep Morton chain
    (box centres at cells 1024 - 2^(10-j): codes with their k high bits set, so that the chain runs through RIGHT children and the left ones pile up on the stack) with 64 identical-code triangles hanging off the deepest cell, every box containing one common line — and a
    ray along that line that misses every triangle: the deepest the stack gets is depth + 1, far below 64, nothing overflows, and the answer is a miss."""
    v, t, (lx, ly) = adversarial_chain_mesh()
    info, aabb, srt, h = oracle.bvh_build(v, t)
    depth = oracle.tree_depth(info)
    assert depth <= 30 + int(np.ceil(np.log2(len(t)))) and depth >= 27          # the chain is there
    rays = oracle.make_rays(np.array([[1.0 - lx, 1.0 - ly, -0.5]], np.float32), np.array([[0.0, 0.0, 1.0]], np.float32))
    r = oracle.trace(info, aabb, v, t, rays, True, True)
    deepest = int(oracle.trace_stack_depth(info, aabb, v, t, rays)[0])
    assert r["hit"][0] == 0 and r["counters"][0, 3] == 0
    assert r["counters"][0, 2] >= 60, r["counters"][0]                          # the ray really goes through (nearly) every leaf box
    assert depth // 2 < deepest <= depth + 1 < 64, (depth, deepest)


def test_behind_the_origin_hits_end_the_search_at_the_highest_slot(oracle, scene_mod):
    """The order-free statement k_trace_closest4 relies on (bvh_trace.hip, round 4), checked against the oracle's bvh_hit on the lego-like mesh WITHOUT any
    hierarchy: among the triangles that (1) Moller-Trumbore accepts with t <= t_min = 0 and (2) whose own leaf box passes aabb_hit against [0, anything > 0] — all of
    it re-stated here in numpy float32 with the reference's operation order — the reference's right-first traversal reports the one with the HIGHEST sorted
    position (slot), with that triangle's t. Rays: from surface points of the mesh, where large triangles' boxes contain the origins of many rays."""
    f = np.float32
    v, t = scene_mod.make_mesh_clustered(13000)
    info, aabb, srt, h = oracle.bvh_build(v, t)
    T = len(t)
    slot_of = np.empty(T, np.int64); slot_of[srt[:, 1]] = np.arange(T)
    eye, rd = scene_mod.camera_rays(30, 30)
    prim0 = oracle.trace(info, aabb, v, t, oracle.make_rays(np.repeat(eye[None], 900, 0), rd), True)
    fg = prim0["hit"] > 0
    rng = np.random.default_rng(4)
    g = rng.normal(size=(int(fg.sum()), 3)).astype(f); g /= np.linalg.norm(g, axis=1, keepdims=True)
    d = (prim0["normal"][fg] + f(0.9) * g).astype(f)
    o = (prim0["pos"][fg] + f(0.01) * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f)
    r = oracle.trace(info, aabb, v, t, oracle.make_rays(o, d), True)
    v0 = v[t[:, 0]]; E1 = (v[t[:, 1]] - v0).astype(f); E2 = (v[t[:, 2]] - v0).astype(f)
    bmin = np.minimum(np.minimum(v[t[:, 0]], v[t[:, 1]]), v[t[:, 2]]); bmax = np.maximum(np.maximum(v[t[:, 0]], v[t[:, 1]]), v[t[:, 2]])
    dot = lambda a, b: (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]
    def cross(a, b):
        return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1).astype(f)
    seen = 0
    with np.errstate(all="ignore"):
        for i in range(len(o)):
            dn = d[i] * (f(1.0) / np.sqrt(f((d[i, 0] * d[i, 0] + d[i, 1] * d[i, 1]) + d[i, 2] * d[i, 2])))          # normalize (helperDi.slang:201)
            dn = dn.astype(f)
            D = np.repeat(dn[None], T, 0)
            P = cross(D, E2); det = dot(E1, P)
            ok = ~((det > f(-1e-15)) & (det < f(1e-15)))
            inv = (f(1) / det).astype(f); Tv = (o[i][None] - v0).astype(f)
            u = (dot(Tv, P) * inv).astype(f); ok &= ~((u < 0) | (u > 1))
            Q = cross(Tv, E1); w = (dot(D, Q) * inv).astype(f); ok &= ~((w < 0) | ((u + w).astype(f) > 1))
            tt = (dot(E2, Q) * inv).astype(f)
            # aabb_hit of the triangle's own box against [t_min = 0, t_max = +inf) (helperDi.slang:149-170)
            tmin = np.zeros(T, f); tmax = np.full(T, np.inf, f); passed = np.ones(T, bool)
            for a in range(3):
                da = dn[a] if dn[a] != 0 else f(1e-6)
                ia = f(1.0) / da
                t0 = ((bmin[:, a] - o[i, a]) * ia).astype(f); t1 = ((bmax[:, a] - o[i, a]) * ia).astype(f)
                if ia < 0: t0, t1 = t1, t0
                tmin = np.where(t0 > tmin, t0, tmin); tmax = np.where(t1 < tmax, t1, tmax)
                passed &= ~(tmax <= tmin)
            cand = ok & passed & (tt <= 0)
            if not cand.any():
                continue
            seen += 1
            best = np.flatnonzero(cand)[np.argmax(slot_of[cand])]
            assert r["hit"][i] == 1 and r["prim"][i] == best and r["t"][i] == tt[best], (i, int(r["prim"][i]), int(best), float(r["t"][i]), float(tt[best]))
    assert seen > 30, seen          # the situation is common on this mesh


def test_negative_t_quirk_is_preserved(oracle):
    """triangle_hit ignores the t interval (helperDi.slang:172-195): a triangle BEHIND the origin whose box contains the origin is a hit."""
    v = np.array([[-1, -1, 0.3], [1, -1.1, -0.3], [0, 1, 0.1], [5, 5, 5], [6, 5, 5.2], [5, 6, 5.1]], np.float32)
    t = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    info, aabb, _, _ = oracle.bvh_build(v, t)
    rays = oracle.make_rays(np.array([[0, 0, 0.2]], np.float32), np.array([[0.05, 0.02, 1]], np.float32))
    r = oracle.trace(info, aabb, v, t, rays, True)
    assert r["hit"][0] == 1 and r["t"][0] < 0 and r["prim"][0] == 0


def test_restir_di_converges_to_quadrature(oracle, scene_mod):
    """E[color] of the ReSTIR-DI estimator (initial + temporal + spatial + final shading, un-denoised) equals the direct-lighting integral
    sum_k f(w_k) Le(w_k) V(w_k) dw_k evaluated by brute-force quadrature with the SAME shading function. Foreground mean within 4 %."""
    F = SmallFrame(oracle, scene_mod, fx=20, fy=16, subdiv=2, ground=4, env_hw=(16, 32), varied=False, rough=0.6)
    O = oracle
    N = F.N
    spp = 192
    out = O.render(F.fx, F.fy, spp, 777, (F.info, F.aabb), F.vert, F.tri, F.env, F.occ, F.normal, F.depth, F.kd, F.rm, F.ray_dir_raw, F.pos, mat=None, want_avg=True)
    est = out["avg_direct"]
    K = 3000
    k = np.arange(K) + 0.5
    z = 1 - 2 * k / K; phi = k * np.pi * (3 - np.sqrt(5)); rr = np.sqrt(1 - z * z)
    dirs = np.stack([rr * np.cos(phi), rr * np.sin(phi), z], 1).astype(np.float32)
    Le = O.env_le(F.tex, F.Wc, F.Hc, np.stack([-dirs[:, 0], dirs[:, 2], dirs[:, 1]], 1).astype(np.float32))      # env_radiance(L) = env_le(ngp_dir(L))
    acc = np.zeros((N, 3), np.float64)
    fg = F.occ > 0.5
    pos = F.pos[fg]
    for j in range(K):
        if Le[j].max() <= 0:
            continue
        L = dirs[j]
        rays = O.make_rays(pos + 0.01 * L, np.repeat(L[None], len(pos), 0))
        hit = O.trace(F.info, F.aabb, F.vert, F.tri, rays, False)["hit"]
        vis = np.zeros(N, np.float32); vis[fg] = 1.0 - hit
        fdir = np.repeat(L[None], N, 0).astype(np.float32); fdist = np.where(vis > 0, 1e6, 0).astype(np.float32)
        fLi = (Le[j][None] * (4 * np.pi / K) * vis[:, None]).astype(np.float32)
        c, _, _ = O.final_shading(F.frame, F.normal, F.kd, F.rm, fdir, fdist, fLi)
        acc[fg] += c[fg]
    a, b = est[fg].mean(0), acc[fg].mean(0)
    np.testing.assert_allclose(a, b, rtol=0.04)
    lum = lambda x: x @ np.array([0.2126, 0.7152, 0.0722])
    cc = np.corrcoef(lum(est[fg]), lum(acc[fg]))[0, 1]
    assert cc > 0.9, cc


def test_eaw_properties(oracle, scene_mod):
    F = SmallFrame(oracle, scene_mod, fx=24, fy=20, subdiv=2, ground=4, env_hw=(8, 16))
    const = np.full((F.N, 3), 0.4, np.float32)
    out = oracle.eaw(F.fx, F.fy, 2, 2.0, 0.1, 0.001, F.occ, const, F.normal, F.pos)
    np.testing.assert_allclose(out, 0.4, rtol=1e-6)
    rng = np.random.default_rng(0)
    col = rng.random((F.N, 3)).astype(np.float32)
    out = oracle.eaw(F.fx, F.fy, 1, 2.0, 0.1, 0.001, F.occ, col, F.normal, F.pos)
    assert np.array_equal(out[F.occ < 0.5], col[F.occ < 0.5])                                  # background copied
    assert out[F.occ > 0.5].min() >= col.min() - 1e-6 and out[F.occ > 0.5].max() <= col.max() + 1e-6   # convex combination


def test_hashgrid_dense_levels_are_trilinear(oracle, scene_mod):
    total = oracle.hashgrid_layout()[0]
    rng = np.random.default_rng(5)
    params = np.zeros(total * 2, np.float32); params[:2 * 17920] = rng.normal(size=2 * 17920).astype(np.float32) * 0.1     # levels 0 and 1
    keep = oracle.Keep(); mn, mx = scene_mod.material_min_max()
    z = np.zeros((32, 32), np.float32)
    mat = oracle.matnet_struct(keep, params, z, z, np.zeros((6, 32), np.float32), (-1, -1, -1), (1, 1, 1), mn, mx)
    x = rng.random((400, 3)).astype(np.float32)
    enc = oracle.hashgrid_encode(mat, x).view(np.float16).astype(np.float64)
    tabs = [(16, 15.0, 0, 4096), (24, 22.156307, 4096, 13824)]
    p16 = oracle.to_f16_bits(params[:2 * 17920]).view(np.float16).astype(np.float64).reshape(-1, 2)
    for lv, (res, scale, off, size) in enumerate(tabs):
        p = x.astype(np.float64) * scale + 0.5
        i0 = np.floor(p).astype(int); fr = p - i0
        acc = np.zeros((400, 2))
        for c in range(8):
            dx, dy, dz = c & 1, (c >> 1) & 1, (c >> 2) & 1
            w = (fr[:, 0] if dx else 1 - fr[:, 0]) * (fr[:, 1] if dy else 1 - fr[:, 1]) * (fr[:, 2] if dz else 1 - fr[:, 2])
            idx = ((i0[:, 0] + dx) + (i0[:, 1] + dy) * res + (i0[:, 2] + dz) * res * res) % size
            acc += w[:, None] * p16[off + idx]
        np.testing.assert_allclose(enc[:, 2 * lv:2 * lv + 2], acc, rtol=0, atol=4e-3 * np.abs(acc).max() + 1e-4)
    assert np.all(enc[:, 4:] == 0)
    out = oracle.matnet(mat, x * 2 - 1)
    np.testing.assert_allclose(out, np.tile((mn + mx) / 2, (400, 1)), atol=1e-7)                 # zero weights -> sigmoid(0) = 0.5


def test_bilateral_denoiser_properties(oracle):
    """nerf/renderutils bilateral denoiser (the --use_bi_de branch of run_restir_di_with_pt): a normalised filter keeps a constant image,
    the fourth channel is the clamped weight sum, and the backward kernel is the exact adjoint of the forward in the colour
    (<bwd(g), c> == <g, fwd(c)>: the depth term's denominator is transposed, c_src/denoising.cu:113)."""
    rng = np.random.default_rng(5)
    fx, fy, sigma = 20, 14, 1.0
    n = fx * fy
    nrm = rng.normal(size=(n, 3)).astype(np.float32); nrm[:, 2] += 3.0
    zdz = np.stack([1.0 + 0.05 * rng.random(n), 0.01 + 0.02 * rng.random(n)], axis=1).astype(np.float32)
    const = np.tile(np.array([[0.3, 0.6, 0.9]], np.float32), (n, 1))
    o = oracle.bilateral(fx, fy, sigma, const, nrm, zdz)
    assert o.shape == (n, 4) and np.all(o[:, 3] >= 1e-4)
    np.testing.assert_allclose(o[:, :3] / o[:, 3:4], const, rtol=2e-6)
    c = rng.random((n, 3)).astype(np.float32); g = rng.normal(size=(n, 4)).astype(np.float32)
    f = oracle.bilateral(fx, fy, sigma, c, nrm, zdz)
    b = oracle.bilateral(fx, fy, sigma, None, nrm, zdz, grad4=g)
    lhs = float((b.astype(np.float64) * c).sum()); rhs = float((g[:, :3].astype(np.float64) * f[:, :3]).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))
    # the centre tap has weight 1, so an isolated bright pixel keeps at least its own contribution
    assert np.all(f[:, 3] >= 0.999)          # (n.n)^128 of a unit normal is 1 up to fp32 rounding


def test_prepare_shading_normal_known_cases(oracle):
    """nerf/renderutils.prepare_shading_normal (numpy restatement): unperturbed front-facing normal is returned unchanged, a back-facing
    geometric normal flips both normals when two-sided, and a grazing smooth normal is bent towards the geometric one (threshold 0.1)."""
    f = np.float32
    pos = np.zeros((1, 3), f); view = np.array([[0, 0, 5]], f)
    up = np.array([[0, 0, 2]], f); tng = np.array([[3, 0, 0]], f); p0 = np.array([[0, 0, 1]], f)
    out = oracle.prepare_shading_normal(pos, view, p0, up, tng, np.array([[0, 0, 1]], f))
    np.testing.assert_allclose(out, [[0, 0, 1]], atol=1e-7)
    out = oracle.prepare_shading_normal(pos, view, p0, -up, tng, np.array([[0, 0, -1]], f), two_sided_shading=True)
    np.testing.assert_allclose(out, [[0, 0, 1]], atol=1e-7)                                  # flipped to face the viewer
    out1 = oracle.prepare_shading_normal(pos, view, p0, -up, tng, np.array([[0, 0, -1]], f), two_sided_shading=False)
    np.testing.assert_allclose(out1, [[0, 0, -1]], atol=1e-7)                                # one-sided: t = 0 -> the geometric normal
    graz = np.array([[1, 0, 0.05]], f)                                                       # smooth normal almost perpendicular to the view
    out2 = oracle.prepare_shading_normal(pos, view, p0, graz, np.array([[0, 1, 0]], f), np.array([[0, 0, 1]], f))
    n = graz / np.linalg.norm(graz); t = (n[0, 2] / 0.1)
    np.testing.assert_allclose(out2, np.array([[0, 0, 1]], f) * (1 - t) + n * t, rtol=1e-5, atol=1e-6)
    # tangent-space perturbation: x picks the tangent, y the bitangent (sign by convention)
    o_gl = oracle.prepare_shading_normal(pos, view, np.array([[0, 1, 1]], f), up, tng, np.array([[0, 0, 1]], f), opengl=True)
    o_dx = oracle.prepare_shading_normal(pos, view, np.array([[0, 1, 1]], f), up, tng, np.array([[0, 0, 1]], f), opengl=False)
    assert o_gl[0, 1] * o_dx[0, 1] < 0 and abs(o_gl[0, 2] - o_dx[0, 2]) < 1e-6


def test_shadow_rays_aimed_at_an_emptied_reservoirs_sample_cannot_be_seen(oracle, scene_mod):
    """What the engine's spatial pass relies on when it does not trace part of its shadow rays (engine.hpp RaySrc::skip_dead; 25 % of them on the lego-like
    mesh): in the REFERENCE's arithmetic (SpatialResampling.slang:262-322, res.slang:173-232) the answer of a shadow ray aimed at the light sample of a
    reservoir with weight 0 only scales terms that are multiplied by that weight, so forcing those answers to "free" or to "occluded" must leave every
    output buffer of a many-sample frame with the same bits.  Scene with large shadowed regions (the lego-like mesh from below its plates), 16 samples so that
    emptied reservoirs travel through temporal and spatial reuse; the control (forcing ALL spatial shadow rays) is what shows the hook bites."""
    v, t = scene_mod.make_mesh_clustered(target_tris=20000, seed=3)
    info, aabb, _, _ = oracle.bvh_build(v, t)
    fx = fy = 72
    eye, rd = scene_mod.camera_rays(fy, fx)
    r = oracle.trace(info, aabb, v, t, oracle.make_rays(np.repeat(eye[None], fx * fy, 0), rd), True)
    occ = r["hit"].astype(np.float32)
    nrm = np.where(occ[:, None] > 0, r["normal"], 0).astype(np.float32)
    depth = np.linalg.norm(r["pos"] - eye, axis=1).astype(np.float32)
    N = fx * fy
    kd = np.full((N, 3), 0.6, np.float32); rm = np.zeros((N, 2), np.float32); rm[:, 0] = 0.5
    env = scene_mod.make_env(64, 128)
    names = ["final_color", "diffuse", "spec", "indirect", "indirect_diff", "indirect_spec"]
    def frame():
        return oracle.render(fx, fy, 16, 777, (info, aabb), v, t, env, occ, nrm, depth, kd, rm, rd, r["pos"], mat=None)
    try:
        oracle.set_dead_ray_override(-1); ref = frame()
        outs, answered = {}, {}
        for mode in (0, 1, 2):
            oracle.set_dead_ray_override(mode); outs[mode] = frame(); answered[mode] = oracle.set_dead_ray_override(-1)
    finally:
        oracle.set_dead_ray_override(-1)
    assert occ.mean() > 0.3 and float(ref["final_color"].mean()) > 0
    for mode in (0, 1):
        for k in names:
            assert np.array_equal(outs[mode][k], ref[k]), (mode, k, int((outs[mode][k] != ref[k]).any(axis=1).sum()))
    # the equality above says something: a good part of the spatial pass's shadow rays is of that kind (the same number in both modes: the reservoirs are the
    # same), and the hook is live — with EVERY spatial shadow ray forced to "occluded" the frame changes
    assert answered[0] == answered[1] and answered[0] > 0.05 * answered[2] > 0, answered
    assert not np.array_equal(outs[2]["final_color"], ref["final_color"])
