"""Shared fixtures for the parity tests: a small synthetic frame whose every stage is produced by the CPU oracle."""
import numpy as np

CONFIG = dict(light_tile_count=128, light_tile_size=1024, neighbor_offset_count=8192)


class SmallFrame:
    """Mesh + G-buffer + env tables built with the oracle only (no GPU). fx x fy pixels, constant or textured materials."""

    def __init__(self, O, S, fx=48, fy=40, subdiv=3, ground=16, env_hw=(32, 64), seed=0, rough=0.45, metal=0.0, varied=True):
        import os
        sweep = int(os.environ.get("MIRRES_TEST_SEED", "0"))      # robustness sweeps: another view, other materials, same tests
        seed += sweep
        self.O, self.fx, self.fy = O, fx, fy
        N = fx * fy
        self.N = N
        self.vert, self.tri = S.make_mesh(subdiv, ground)
        self.info, self.aabb, self.sorted, self.height = O.bvh_build(self.vert, self.tri)
        eye, rd = S.camera_rays(fy, fx, 30.0 + 47.0 * sweep, 30.0 - 7.0 * (sweep % 5)) if sweep else S.camera_rays(fy, fx)
        self.eye = eye
        self.ray_dir_raw = rd
        rays = O.make_rays(np.repeat(eye[None], N, 0), rd)
        r = O.trace(self.info, self.aabb, self.vert, self.tri, rays, True)
        self.occ = r["hit"].astype(np.float32)
        self.pos = r["pos"].copy()
        self.normal = np.where(self.occ[:, None] > 0, r["normal"], 0).astype(np.float32)
        self.depth = np.linalg.norm(self.pos - eye, axis=1).astype(np.float32)
        rng = np.random.default_rng(seed)
        if varied:
            self.kd = (0.25 + 0.6 * rng.random((N, 3))).astype(np.float32)
            self.rm = np.stack([0.15 + 0.7 * rng.random(N), metal + 0.3 * rng.random(N)], 1).astype(np.float32)
        else:
            self.kd = np.full((N, 3), 0.6, np.float32)
            self.rm = np.stack([np.full(N, rough), np.full(N, metal)], 1).astype(np.float32)
        self.env = S.make_env(env_hw[0], env_hw[1], sun=25.0)
        self.Hc, self.Wc = env_hw
        self.tex = O.flip_env(self.env)
        self.tables = O.make_sampleable(self.tex, self.Wc, self.Hc)
        # derived maps of restir_di_with_pt (:279-287) and run_restir_di_with_pt (:484-486)
        self.ray_dir = (rd / np.maximum(np.linalg.norm(rd, axis=1, keepdims=True), 1e-6)).astype(np.float32)
        self.normal_depth = np.concatenate([self.normal, self.depth[:, None]], 1).astype(np.float32)
        k, m, r_ = self.kd, self.rm[:, 1], self.rm[:, 0]
        b0 = (k[:, 0] * np.float32(0.2126) + k[:, 1] * np.float32(0.7152)) + k[:, 2] * np.float32(0.0722)
        b1 = (m * np.float32(0.2126) + m * np.float32(0.7152)) + m * np.float32(0.0722)
        a = np.clip(r_, np.float32(0.01), np.float32(1.0))
        self.brdf = np.stack([b0, b1, a * a], 1).astype(np.float32)
        self.noff = O.neighbor_offsets(8192)
        self.keep = O.Keep()
        self.frame = O.make_frame(self.keep, fx, fy, self.occ, self.pos, self.normal_depth, self.brdf, self.ray_dir, (self.info, self.aabb), self.vert,
                                  self.tri, self.tex, self.Wc, self.Hc, self.tables)


def match_fraction(a, b, rtol=1e-4, atol=1e-6):
    """Fraction of rows whose every component agrees within tolerance."""
    a = np.asarray(a, np.float64).reshape(len(a), -1); b = np.asarray(b, np.float64).reshape(len(b), -1)
    ok = np.all(np.abs(a - b) <= atol + rtol * np.abs(b), axis=1)
    import os
    rep = os.environ.get("MIRRES_PARITY_REPORT")
    if rep:
        import inspect
        fr = inspect.stack()[1]
        with open(rep, "a") as f:
            f.write("%s:%d %s: %d of %d rows outside rtol %g / atol %g; bit-equal rows %.5f; max abs diff %.3e\n" % (
                os.path.basename(fr.filename), fr.lineno, fr.function, int((~ok).sum()), len(ok), rtol, atol, float(np.all(a == b, axis=1).mean()), float(np.abs(a - b).max()) if a.size else 0.0))
    return float(ok.mean()), ok


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def torch_material_field(O, params_f32, w0, w1, w2, aabb_min, aabb_max, mn, mx, pos, dtype=None, table=None):
    """Plain-torch reference of MLPTexture3D.sample (render_helper.py:93-104) with autograd: position normalisation + clamp, the hash-grid encoding
    (tcnn's published algorithm: `fmaf(scale, x, 0.5)`, floor, 8-corner trilinear weights, dense / coherent-prime-hash index; the table holds the
    fp16-rounded parameters but the interpolation is carried out in `dtype`, not fp16), the bias-free 32-32-32-6 ReLU MLP, sigmoid and range.
    `pos` may require grad; everything stays on pos.device.  The level layout comes from the oracle (`hashgrid_layout`).  `table`: the (fp16-valued) parameter
    table as a flat tensor that may require grad (then params_f32 is not read) — for the table's own gradient."""
    import torch
    dtype = dtype or torch.float64
    dev = pos.device
    total, off, res, sc = O.hashgrid_layout()
    tab = table.reshape(-1, 2) if table is not None else torch.from_numpy(O.to_f16_bits(params_f32).view(np.float16).astype(np.float64).reshape(-1, 2)).to(dev, dtype)
    lo = torch.tensor(aabb_min, dtype=dtype, device=dev); hi = torch.tensor(aabb_max, dtype=dtype, device=dev)
    x = torch.clamp((pos.to(dtype) - lo) / (hi - lo), 0, 1)
    feats = []
    for lv in range(16):
        size = int(off[lv + 1]) - int(off[lv]); r = int(res[lv])
        q = float(sc[lv]) * x + 0.5
        cell = torch.floor(q.detach()).to(torch.int64)
        w = q - cell.to(dtype)
        acc = 0
        for idx in range(8):
            c = [cell[:, d] + ((idx >> d) & 1) for d in range(3)]
            wt = 1
            for d in range(3):
                wt = wt * (w[:, d] if (idx >> d) & 1 else 1 - w[:, d])
            if r * r * r <= size:          # dense level (the stride never exceeds the level's size)
                index = c[0] + c[1] * r + c[2] * r * r
            else:
                m32 = 0xFFFFFFFF
                index = ((c[0] * 1) & m32) ^ ((c[1] * 2654435761) & m32) ^ ((c[2] * 805459861) & m32)
            index = index % size + int(off[lv])
            acc = acc + wt[:, None] * tab[index]
        feats.append(acc)
    a = torch.cat(feats, dim=1)
    W = [torch.as_tensor(t).to(dev, dtype) for t in (w0, w1, w2)]
    h = torch.relu(a @ W[0].T); h = torch.relu(h @ W[1].T); z = h @ W[2].T
    mn_t = torch.tensor(np.asarray(mn, np.float64), device=dev, dtype=dtype); mx_t = torch.tensor(np.asarray(mx, np.float64), device=dev, dtype=dtype)
    return torch.sigmoid(z) * (mx_t - mn_t) + mn_t


def antialias_ref(color, rast, pos, tri, opp, H, W):
    """Plain-numpy (float64, pixel loops) statement of the published dr.antialias algorithm as csrc/antialias.hip documents it — independent code, same
    rules: for every horizontally / vertically adjacent pixel pair with different triangle ids take the nearer triangle (background defers), find its
    silhouette edges (boundary edge, or the neighbour across lies on the same side in screen space), the nearest crossing u in [0, 1] of the segment
    between the two pixel centres, alpha = u - 1/2, and blend: alpha > 0 -> out[p1] += alpha (in[p0] - in[p1]); alpha < 0 -> out[p0] += -alpha (in[p1] - in[p0]).
    Returns (out [H*W, C], list of (p0, p1, alpha, va, vb))."""
    color = np.asarray(color, np.float64); rast = np.asarray(rast, np.float64); pos = np.asarray(pos, np.float64)
    out = color.copy()
    pairs = []

    def proj(v, cx, cy):
        x, y, z, w = pos[v]
        if not w > 0:
            return None
        return np.array([(x / w + 1) * 0.5 * W - cx, (y / w + 1) * 0.5 * H - cy])

    for lo in range(H * W):
        for axis, hi in ((0, lo + 1 if (lo % W) + 1 < W else -1), (1, lo + W if lo + W < H * W else -1)):
            if hi < 0:
                continue
            t0, t1 = int(rast[lo, 3]) - 1, int(rast[hi, 3]) - 1
            if t0 == t1:
                continue
            first = t0 >= 0
            if t0 >= 0 and t1 >= 0:
                first = not (rast[hi, 2] < rast[lo, 2])
            t = t0 if first else t1
            p0, p1 = (lo, hi) if first else (hi, lo)
            s = 1.0 if first else -1.0
            cx, cy = (p0 % W) + 0.5, (p0 // W) + 0.5
            P = [proj(int(v), cx, cy) for v in tri[t]]
            if any(q is None for q in P):
                continue
            best = None
            for k in range(3):
                a, b, c = P[k], P[(k + 1) % 3], P[(k + 2) % 3]
                o = int(opp[t, k])
                if o >= 0:
                    q = proj(o, cx, cy)
                    if q is None:
                        continue
                    e = b - a
                    sc = e[0] * (c[1] - a[1]) - e[1] * (c[0] - a[0]); so = e[0] * (q[1] - a[1]) - e[1] * (q[0] - a[0])
                    if (sc > 0) != (so > 0):
                        continue
                ad, ao, bd, bo = a[axis], a[1 - axis], b[axis], b[1 - axis]
                if (ao > 0) == (bo > 0):
                    continue
                u = s * (ad * bo - bd * ao) / (bo - ao)
                if not (0 <= u <= 1):
                    continue
                if best is None or u < best[0]:
                    best = (u, int(tri[t][k]), int(tri[t][(k + 1) % 3]))
            if best is None:
                continue
            alpha = best[0] - 0.5
            if alpha > 0:
                out[p1] += alpha * (color[p0] - color[p1])
            elif alpha < 0:
                out[p0] += -alpha * (color[p1] - color[p0])
            pairs.append((p0, p1, alpha, best[1], best[2]))
    return out, pairs


def pixel_parity(got, ref, what, tol=1e-3, min_frac=1.0):
    """The north-star bar: every pixel (row) of `got` within `tol` (absolute, per channel) of the oracle's `ref`.  Since round 3 the HIP product and
    the oracle evaluate their transcendental functions with the same arithmetic (include/mirres_fmath.h), so discrete sampler decisions cannot
    diverge through them and the default is ALL pixels; a caller that compares through the MFMA material field (hi / lo split operands, 3e-6 from
    the fp32 chain) states its own `min_frac` and why.  Returns (fraction within tol, number of offenders, max abs error).
    MIRRES_PARITY_REPORT=<file> appends one line per call (what the GPU run observed, pass or fail)."""
    import os
    g = np.asarray(got, np.float64).reshape(len(got), -1); r = np.asarray(ref, np.float64).reshape(len(ref), -1)
    err = np.abs(g - r).max(axis=1)
    bad = ~(err <= tol)
    frac = 1.0 - float(bad.mean())
    mx = float(np.nanmax(err)) if len(err) else 0.0
    rep = os.environ.get("MIRRES_PARITY_REPORT")
    if rep:
        with open(rep, "a") as f:
            f.write("%s: %d of %d rows beyond %g (max abs err %.3e, exact-equal rows %.4f)\n" % (what, int(bad.sum()), len(err), tol, mx, float((err == 0).mean())))
    assert frac >= min_frac, "%s: %d of %d pixels beyond %g (max abs err %.3e)" % (what, int(bad.sum()), len(err), tol, mx)
    return frac, int(bad.sum()), mx


def same_bits(got, ref, what):
    """Bit-for-bit equality of two float arrays (NaNs compare by payload): what the shared arithmetic of product and oracle buys for everything that does not
    pass through the MFMA material field."""
    import os
    g = np.ascontiguousarray(np.asarray(got, np.float32)).reshape(-1); r = np.ascontiguousarray(np.asarray(ref, np.float32)).reshape(-1)
    assert g.shape == r.shape, (what, g.shape, r.shape)
    bad = g.view(np.uint32) != r.view(np.uint32)
    rep = os.environ.get("MIRRES_PARITY_REPORT")
    if rep:
        with open(rep, "a") as f:
            f.write("%s: %d of %d values differ in bits (max abs diff %.3e)\n" % (what, int(bad.sum()), bad.size, float(np.nanmax(np.abs(g - r))) if bad.any() else 0.0))
    assert not bad.any(), "%s: %d of %d values differ (first at %d: %r vs %r, max abs diff %.3e)" % (
        what, int(bad.sum()), bad.size, int(np.argmax(bad)), g[np.argmax(bad)], r[np.argmax(bad)], float(np.nanmax(np.abs(g - r))))


def rasterize_ref(pos_clip, tri, H, W):
    """Software rasteriser in numpy float64 (test reference for raster.rasterize, written from nvdiffrast's published output contract): for every pixel centre
    NDC ((2 ix + 1) / W - 1, (2 iy + 1) / H - 1) the front-most triangle (smallest z/w in [-1, 1]) among those whose projected 2-D triangle contains it;
    record (u, v, z/w, id + 1) with perspective-correct barycentrics u, v = weights of v0, v1.  Also returns `margin`: the smallest absolute screen-space
    barycentric of the winner and the z/w gap to the runner-up (pixels with tiny margins may legitimately resolve differently in float32)."""
    p = np.asarray(pos_clip, np.float64); t = np.asarray(tri, np.int64)
    ndc = p[:, :3] / p[:, 3:4]; wv = p[:, 3]
    xs = (2 * np.arange(W) + 1) / W - 1; ys = (2 * np.arange(H) + 1) / H - 1
    px, py = np.meshgrid(xs, ys); px = px.ravel(); py = py.ravel()
    n = H * W
    best_z = np.full(n, np.inf); second_z = np.full(n, np.inf); rast = np.zeros((n, 4)); edge = np.full(n, np.inf)
    a, b, c = ndc[t[:, 0]], ndc[t[:, 1]], ndc[t[:, 2]]
    area = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    for k in range(len(t)):
        if area[k] == 0 or (wv[t[k]] <= 0).any():
            continue
        w0 = ((b[k, 0] - px) * (c[k, 1] - py) - (b[k, 1] - py) * (c[k, 0] - px)) / area[k]
        w1 = ((c[k, 0] - px) * (a[k, 1] - py) - (c[k, 1] - py) * (a[k, 0] - px)) / area[k]
        w2 = 1 - w0 - w1
        inside = (w0 >= 0) & (w1 >= 0) & (w2 >= 0)
        if not inside.any():
            continue
        z = w0 * a[k, 2] + w1 * b[k, 2] + w2 * c[k, 2]
        ok = inside & (z >= -1) & (z <= 1)
        closer = ok & (z < best_z)
        second_z = np.where(closer, best_z, np.where(ok & (z < second_z), z, second_z))
        q0, q1, q2 = w0 / wv[t[k, 0]], w1 / wv[t[k, 1]], w2 / wv[t[k, 2]]
        s = q0 + q1 + q2
        rast[closer] = np.stack([q0 / s, q1 / s, z, np.full(n, k + 1.0)], 1)[closer]
        edge = np.where(closer, np.minimum(np.minimum(w0, w1), w2), edge)
        best_z = np.where(closer, z, best_z)
    with np.errstate(invalid="ignore"):
        gap = second_z - best_z
    return rast, edge, gap


def rasterize_ref_homogeneous(pos_clip, tri, H, W):
    """The same contract by 2-D homogeneous rasterisation (Olano & Greer 1997) in float64, clipping included: for clip-space vertices (x_i, y_i, w_i) the
    columns of the inverse of [[x0 y0 w0], [x1 y1 w1], [x2 y2 w2]] are the coefficients of three functions a_i(x, y), linear in NDC, with
    a_i / (a_0 + a_1 + a_2) = the perspective-correct barycentric of vertex i and a_0 + a_1 + a_2 = 1 / w; a pixel centre is covered where all three have the
    sign of 1 / w > 0, whether or not some vertices lie behind the eye.  Fragments with z / w outside [-1, 1] are discarded one by one — for a point-sampled
    rasteriser that is what clipping against the near and far planes does.  Returns (rast [n,4], rast_db [n,4] = analytic (du/dX, du/dY, dv/dX, dv/dY) per
    pixel step, margin [n] = the smallest barycentric of the winner, gap [n] = z/w distance to the runner-up or to the clip planes)."""
    p = np.asarray(pos_clip, np.float64); t = np.asarray(tri, np.int64)
    xs = (2 * np.arange(W) + 1) / W - 1; ys = (2 * np.arange(H) + 1) / H - 1
    px, py = np.meshgrid(xs, ys); px = px.ravel(); py = py.ravel()
    n = H * W
    best_z = np.full(n, np.inf); second_z = np.full(n, np.inf); rast = np.zeros((n, 4)); db = np.zeros((n, 4)); edge = np.full(n, np.inf); plane = np.full(n, np.inf)
    for k in range(len(t)):
        V = p[t[k]]                                              # rows = vertices, columns x y z w
        Mx = V[:, [0, 1, 3]]
        det = np.linalg.det(Mx)
        if abs(det) < 1e-300:
            continue
        Mi = np.linalg.inv(Mx)                                   # (x, y, 1) . Mi[:, i] = a_i
        a = px[:, None] * Mi[0][None] + py[:, None] * Mi[1][None] + Mi[2][None]
        s = a.sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            b = a / s[:, None]
        inside = (s > 0) & (b >= 0).all(1)
        if not inside.any():
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            z = (b @ V[:, 2]) / (b @ V[:, 3])
        frag = inside & np.isfinite(z)
        plane = np.where(frag, np.minimum(plane, np.minimum(np.abs(z + 1), np.abs(z - 1))), plane)
        ok = frag & (z >= -1) & (z <= 1)
        closer = ok & (z < best_z)
        second_z = np.where(closer, best_z, np.where(ok & (z < second_z), z, second_z))
        sx, sy = Mi[0].sum(), Mi[1].sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            dbx = (Mi[0][None] - b * sx) / s[:, None] * (2.0 / W)     # d b_i / dX
            dby = (Mi[1][None] - b * sy) / s[:, None] * (2.0 / H)
        rec = np.stack([b[:, 0], b[:, 1], z, np.full(n, k + 1.0)], 1)
        rast[closer] = rec[closer]
        db[closer] = np.stack([dbx[:, 0], dby[:, 0], dbx[:, 1], dby[:, 1]], 1)[closer]
        edge = np.where(closer, b.min(1), edge)
        best_z = np.where(closer, z, best_z)
    with np.errstate(invalid="ignore"):
        gap = np.where(np.isfinite(best_z), np.minimum(second_z - best_z, plane), plane)
    return rast, db, edge, gap


def spawn_ranks(fn, args, nprocs, limit=300):
    """torch.multiprocessing.spawn with a deadline: a rendezvous or an exchange that never completes fails the test instead of hanging the suite (the processes
    started here are killed by their own handles)."""
    import time
    import pytest
    import torch.multiprocessing as mp
    pc = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.time() + limit
    while not pc.join(timeout=5):
        if time.time() > deadline:
            for pr in pc.processes:
                if pr.is_alive():
                    pr.kill()
            pytest.fail("%d ranks of %s did not finish within %d s" % (nprocs, getattr(fn, "__name__", "worker"), limit))

