"""Synthetic inputs for the hot path: no datasets or checkpoints exist offline.

Mesh   : noise-displaced icosphere + tilted, tessellated ground quad (no axis-aligned triangle, so the
         reference's zero-thickness-box quirk, helperDi.slang:165, never triggers by construction).
Camera : NeRF-blender convention restated from nerf/utils.py:408-416 (dir = ((i+.5-cx)/fx, -(j+.5-cy)/fy, -1) @ R^T).
Env    : 256x512 procedural sky (vertical gradient + sun lobe), lat-long, float32.
Pure numpy; deterministic for a given seed. Used by tests/ and bench.py.
"""
import numpy as np


def icosphere(subdiv):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(subdiv):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        es = np.sort(e, axis=1)
        key = es[:, 0] * (len(v) + 1) + es[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a = uniq // (len(v) + 1)
        b = uniq % (len(v) + 1)
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid], axis=0)
        n = len(f)
        m01, m12, m20 = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)], axis=0)
    return v, f


def _value_noise(p, seed, res=16):
    rng = np.random.default_rng(seed)
    tab = rng.random((res, res, res))
    q = (p * 0.5 + 0.5) * res
    i0 = np.floor(q).astype(np.int64)
    fr = q - i0
    fr = fr * fr * (3 - 2 * fr)
    out = np.zeros(len(p))
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                w = (fr[:, 0] if dx else 1 - fr[:, 0]) * (fr[:, 1] if dy else 1 - fr[:, 1]) * (fr[:, 2] if dz else 1 - fr[:, 2])
                out += w * tab[(i0[:, 0] + dx) % res, (i0[:, 1] + dy) % res, (i0[:, 2] + dz) % res]
    return out


def make_mesh(subdiv=3, ground_res=16, seed=0, radius=0.55):
    """Returns (verts f32[V,3], tris i32[T,3]). subdiv=7, ground_res=64 is the BASELINE config-2 mesh (T=335,872)."""
    v, f = icosphere(subdiv)
    disp = np.zeros(len(v))
    amp, freq = 0.12, 1.7
    for o in range(3):
        disp += amp * (_value_noise(np.clip(v * freq * 0.5, -0.999, 0.999) if o == 0 else np.mod(v * freq * 0.5 + 1, 2) - 1, seed + o) - 0.5)
        amp *= 0.5
        freq *= 2.1
    v = v * (radius * (1.0 + disp))[:, None]
    v[:, 2] += 0.05
    # tilted ground (7 deg about x and about z... rotations chosen so no edge is axis-aligned)
    g = np.linspace(-1.0, 1.0, ground_res + 1)
    gx, gy = np.meshgrid(g, g, indexing="xy")
    gv = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, -0.62)], axis=1)
    a = np.deg2rad(7.0)
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    gv = gv @ (Rz @ Rx).T * 0.8
    idx = np.arange((ground_res + 1) ** 2).reshape(ground_res + 1, ground_res + 1)
    q00, q01, q10, q11 = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    gf = np.concatenate([np.stack([q00, q01, q11], 1), np.stack([q00, q11, q10], 1)], axis=0) + len(v)
    verts = np.concatenate([v, gv], axis=0).astype(np.float32)
    tris = np.concatenate([f, gf], axis=0).astype(np.int32)
    return np.ascontiguousarray(verts), np.ascontiguousarray(tris)


def camera_rays(H, W, azimuth_deg=30.0, elevation_deg=30.0, radius=3.2, camera_angle_x=0.6911):
    """NeRF-blender camera (z-up world, camera looks down -z). Returns rays_o f32[3], rays_d f32[H*W,3] (unnormalised)."""
    az, el = np.deg2rad(azimuth_deg), np.deg2rad(elevation_deg)
    eye = radius * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    cup = np.cross(right, fwd)
    R = np.stack([right, cup, -fwd], axis=1)  # camera-to-world
    focal = 0.5 * W / np.tan(0.5 * camera_angle_x)
    i, j = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="xy")
    d = np.stack([(i + 0.5 - W * 0.5) / focal, -(j + 0.5 - H * 0.5) / focal, -np.ones_like(i)], axis=-1).reshape(-1, 3)
    rays_d = d @ R.T
    return eye.astype(np.float32), np.ascontiguousarray(rays_d.astype(np.float32))


def make_env(Hc=256, Wc=512, seed=0, sun=50.0):
    """Lat-long sky [Hc,Wc,3] f32: gradient 0.2 -> 0.8 plus a ~5 deg sun lobe at 40 deg elevation."""
    rng = np.random.default_rng(seed)
    v = (np.arange(Hc) + 0.5) / Hc
    u = (np.arange(Wc) + 0.5) / Wc
    grad = 0.2 + 0.6 * (1.0 - v)
    env = np.repeat(grad[:, None, None], Wc, axis=1) * np.array([0.85, 0.95, 1.1])[None, None, :]
    theta = np.pi * v[:, None]
    phi = 2 * np.pi * u[None, :]
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) * np.ones_like(phi), np.sin(theta) * np.sin(phi)], -1)
    saz = rng.uniform(0, 2 * np.pi)
    sel = np.deg2rad(40.0)
    sd = np.array([np.cos(sel) * np.cos(saz), np.sin(sel), np.cos(sel) * np.sin(saz)])
    ang = np.arccos(np.clip(d @ sd, -1, 1))
    env = env + (sun * np.exp(-0.5 * (ang / np.deg2rad(2.5)) ** 2))[..., None] * np.array([1.0, 0.95, 0.85])
    return np.ascontiguousarray(env.astype(np.float32))


def make_matnet_params(seed=0, scale=1e-4 * 1e3, n_params=12599920):
    """Seeded material-field weights: hash-grid params U(-1e-4,1e-4) x1e3 (so outputs are non-trivial)
    and kaiming-uniform(relu) bias-free Linear weights [32,32],[32,32],[6,32]."""
    rng = np.random.default_rng(seed)
    params = (rng.random(n_params, dtype=np.float32) * 2 - 1) * np.float32(scale)
    def kaiming(out_f, in_f):
        bound = np.sqrt(2.0) * np.sqrt(3.0 / in_f)
        return ((rng.random((out_f, in_f), dtype=np.float32) * 2 - 1) * np.float32(bound)).astype(np.float32)
    return params.astype(np.float32), kaiming(32, 32), kaiming(32, 32), kaiming(6, 32)


# main.py:167-170 / network.py:119-125 : min=(0,0,0, 0,roughness_min,0)  max=(1,1,1, 0,1,me_max)
def material_min_max(roughness_min=0.08, me_max=0.0):
    return (np.array([0, 0, 0, 0, roughness_min, 0], np.float32), np.array([1, 1, 1, 0, 1, me_max], np.float32))


# ---------------------------------------------------------------- a lego-LIKE workload ("What's missing" 1)
# BASELINE's scene is TensoIR-lego: marching cubes of a density field + decimation to 3e5 triangles (main.py:133, configs/tensoir_synthetic/lego.txt).
# No such mesh exists offline, and the icosphere of make_mesh() is the LBVH's best case (uniform tessellation, depth complexity 1-2). This builder
# makes the features that move traversal cost: a brick assembly with studs (many small cylinders), hollow bricks with inner tubes (cavities), thin
# plates a few hundredths apart (near occluders), a slatted grille, wheels and track links (clusters of tiny triangles) next to faces tessellated
# from 1 x 1 to 48 x 48 cells (triangle areas spread > 1000 : 1), all turned out of the axes by a fixed rotation — plus a few EXACTLY axis-aligned
# plates, whose zero-thickness leaf boxes the reference's slab test can never enter (helperDi.slang:165).
class _MeshBuf:
    def __init__(self):
        self.v, self.f, self.nv = [], [], 0

    def add(self, v, f):
        self.v.append(np.asarray(v, np.float64)); self.f.append(np.asarray(f, np.int64) + self.nv); self.nv += len(v)

    def grid(self, p0, du, dv, nu, nv, bump=None):
        """Parallelogram p0 + s du + t dv as nu x nv cells (2 nu nv triangles), counter-clockwise seen against du x dv."""
        nu, nv = max(1, int(nu)), max(1, int(nv))
        s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
        p = np.asarray(p0, np.float64)[None] + s.reshape(-1, 1) * np.asarray(du, np.float64)[None] + t.reshape(-1, 1) * np.asarray(dv, np.float64)[None]
        if bump is not None and nu * nv >= 16:          # marching-cubes surfaces are not flat: a smooth normal displacement on the finely tessellated faces
            n = np.cross(du, dv); n = n / max(np.linalg.norm(n), 1e-30)
            inner = ((s > 0) & (s < 1) & (t > 0) & (t < 1)).reshape(-1)     # borders stay put so that neighbouring faces still meet
            p = p + (inner * bump[0] * (_value_noise(np.clip(p * 0.9, -0.999, 0.999), bump[1], res=32) - 0.5))[:, None] * n[None]
        idx = np.arange((nu + 1) * (nv + 1)).reshape(nv + 1, nu + 1)
        a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
        self.add(p, np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0))

    def box(self, c, size, res=(1, 1, 1), skip=(), bump=None):
        """Axis-parallel box (model space) with per-axis cell counts; `skip` names faces left open: '+x', '-z', ..."""
        c = np.asarray(c, np.float64); h = 0.5 * np.asarray(size, np.float64)
        ex, ey, ez = np.array([2 * h[0], 0, 0]), np.array([0, 2 * h[1], 0]), np.array([0, 0, 2 * h[2]])
        lo = c - h
        faces = {"-z": (lo, ey, ex, res[1], res[0]), "+z": (lo + ez, ex, ey, res[0], res[1]), "-y": (lo, ex, ez, res[0], res[2]),
                 "+y": (lo + ey, ez, ex, res[2], res[0]), "-x": (lo, ez, ey, res[2], res[1]), "+x": (lo + ex, ey, ez, res[1], res[2])}
        for name, (p0, du, dv, nu, nv) in faces.items():
            if name not in skip:
                self.grid(p0, du, dv, nu, nv, bump)

    def cylinder(self, base, axis, radius, height, seg, rings=1, cap0=False, cap1=True, inward=False):
        """Cylinder side (seg x rings quads) around `axis` (0/1/2) from `base`, with optional fan caps (seg slivers each)."""
        seg, rings = max(5, int(seg)), max(1, int(rings))
        th = np.linspace(0, 2 * np.pi, seg, endpoint=False)
        u, w = [(1, 2), (2, 0), (0, 1)][axis]
        ring = np.zeros((seg, 3)); ring[:, u] = np.cos(th) * radius; ring[:, w] = np.sin(th) * radius
        ax = np.zeros(3); ax[axis] = 1.0
        pts = np.concatenate([np.asarray(base, np.float64)[None] + ring + ax[None] * (height * k / rings) for k in range(rings + 1)], 0)
        i = np.arange(seg); j = (i + 1) % seg
        f = []
        for k in range(rings):
            a, b, c, d = k * seg + i, k * seg + j, (k + 1) * seg + j, (k + 1) * seg + i
            f += [np.stack([a, b, c], 1), np.stack([a, c, d], 1)]
        f = np.concatenate(f, 0)
        if inward:
            f = f[:, ::-1]
        self.add(pts, f)
        for on, k in ((cap0, 0), (cap1, rings)):
            if on:
                ctr = np.asarray(base, np.float64) + ax * (height * k / rings)
                p = np.concatenate([pts[k * seg:(k + 1) * seg], ctr[None]], 0)
                fan = np.stack([i, j, np.full(seg, seg)], 1)
                self.add(p, fan if k else fan[:, ::-1])


def _clustered_parts(B, s, rng):
    """The assembly in model space (z up, roughly [-1, 1] x [-0.75, 0.75] x [-0.6, 0.62]); s scales every tessellation count."""
    r = lambda n: max(1, int(round(n * s)))
    seg = lambda n: max(6, int(round(n * min(1.0, 0.35 + 0.65 * s))))
    bump = (0.004, 7)
    def studs(x0, x1, y0, y1, z, pitch=0.1, rad=0.03, h=0.024):
        xs = np.arange(x0 + pitch / 2, x1, pitch); ys = np.arange(y0 + pitch / 2, y1, pitch)
        for x in xs:
            for y in ys:
                B.cylinder((x, y, z), 2, rad, h, seg(16), 1, cap0=False, cap1=True)
    def brick(c, size, res, stud=True, skip=(), tubes=False):
        B.box(c, size, (r(res[0]), r(res[1]), r(res[2])), skip=skip, bump=bump)
        c = np.asarray(c, float); h = 0.5 * np.asarray(size, float)
        if stud:
            studs(c[0] - h[0], c[0] + h[0], c[1] - h[1], c[1] + h[1], c[2] + h[2])
        if tubes:   # a hollow brick seen from its open side: inner walls one plate thickness inside, and the tubes between the studs' undersides
            t = 0.02
            B.box(c, (size[0] - 2 * t, size[1] - 2 * t, size[2] - 2 * t), (r(res[0]), r(res[1]), r(res[2])), skip=skip, bump=None)
            for x in np.arange(c[0] - h[0] + 0.1, c[0] + h[0] - 0.05, 0.1):
                B.cylinder((x, c[1], c[2] - h[2] + t), 2, 0.032, size[2] - 2 * t, seg(14), r(3), cap0=False, cap1=False)
                B.cylinder((x, c[1], c[2] - h[2] + t), 2, 0.024, size[2] - 2 * t, seg(14), r(3), cap0=False, cap1=False, inward=True)
    # base plate: huge coarse underside and rim, finely tessellated top between the studs
    B.box((0, 0, -0.56), (2.0, 1.9, 0.06), (r(40), r(38), 1), skip=("-z",), bump=bump)
    B.grid((-1.0, -0.95, -0.59), (0, 1.9, 0), (2.0, 0, 0), 1, 1)                     # two triangles of area 1.9 each
    studs(-1.0, 1.0, -0.95, 0.95, -0.53)
    # chassis and body: bricks of assorted sizes and tessellations (decimation keeps flat faces coarse and detailed regions fine)
    brick((-0.05, 0, -0.43), (1.5, 0.8, 0.2), (24, 12, 4), stud=False)
    brick((-0.35, 0, -0.23), (0.8, 0.7, 0.2), (40, 36, 10))
    brick((0.35, 0.0, -0.26), (0.5, 0.6, 0.14), (6, 6, 2))
    brick((-0.45, 0.0, -0.03), (0.5, 0.5, 0.2), (48, 48, 20), skip=("+y",), tubes=True)      # hollow brick open towards +y: an interior cavity
    brick((0.3, -0.18, -0.12), (0.3, 0.2, 0.14), (3, 2, 1))
    brick((0.3, 0.18, -0.12), (0.3, 0.2, 0.14), (30, 20, 14))
    # cab: thin pillars, a roof plate with studs, a seat
    for sx in (-0.68, -0.24):
        for sy in (-0.22, 0.22):
            B.box((sx, sy, 0.27), (0.035, 0.035, 0.4), (r(2), r(2), r(24)), bump=None)
    brick((-0.46, 0, 0.49), (0.56, 0.56, 0.04), (28, 28, 2))
    brick((-0.5, 0, 0.12), (0.2, 0.3, 0.1), (10, 14, 5), stud=False)
    # exhaust and levers: thin tall cylinders
    B.cylinder((0.42, 0.2, -0.05), 2, 0.035, 0.45, seg(20), r(30), cap1=True)
    B.cylinder((0.42, 0.2, 0.40), 2, 0.05, 0.06, seg(20), r(4), cap0=True, cap1=True)
    for k in range(3):
        B.cylinder((-0.3 + 0.05 * k, -0.1 + 0.1 * k, 0.07), 2, 0.008, 0.2, seg(8), r(10), cap1=True)
    # blade: three thin plates 0.02 apart (near occluders), finely and coarsely tessellated in turn, on two slanted arms
    for k, (res_x, res_z) in enumerate(((64, 24), (2, 1), (32, 12))):
        B.box((0.86 + 0.035 * k, 0, -0.3), (0.015, 1.3, 0.36), (1, r(res_x), r(res_z)), bump=None)
    for sy in (-0.45, 0.45):
        p0 = np.array([0.5, sy - 0.03, -0.42]); L = np.array([0.36, 0.0, 0.1]); Wd = np.array([0, 0.06, 0]); Hh = np.cross(L, Wd); Hh = Hh / np.linalg.norm(Hh) * 0.05
        for (a, du, dv, nu, nv) in ((p0, L, Wd, 12, 2), (p0 + Hh, Wd, L, 2, 12), (p0, Hh, L, 2, 12), (p0 + Wd, L, Hh, 12, 2)):
            B.grid(a, du, dv, r(nu), r(nv))
    # grille: thin slats 0.012 apart in front of a back plate — many near occluders and tiny triangles in one place
    for k in range(36):
        B.box((0.615, -0.21 + 0.012 * k, -0.04), (0.05, 0.004, 0.26), (r(3), 1, r(14)), bump=None)
    B.box((0.57, 0, -0.04), (0.01, 0.46, 0.28), (1, r(20), r(12)), bump=None)
    # tracks: wheels (cylinder + hub tube) and links on both sides
    for sy in (-0.62, 0.62):
        for k in range(9):
            x = -0.8 + 0.2 * k
            B.cylinder((x, sy - 0.06, -0.4), 1, 0.085, 0.12, seg(28), r(4), cap0=True, cap1=True)
            B.cylinder((x, sy - 0.07, -0.4), 1, 0.03, 0.14, seg(12), r(2), cap0=True, cap1=True)
        for k in range(44):
            x = -0.92 + 0.042 * k
            B.box((x, sy, -0.3), (0.034, 0.15, 0.016), (r(2), r(8), 1), bump=None)
            B.box((x, sy, -0.5), (0.034, 0.15, 0.016), (r(2), r(8), 1), bump=None)
    # two towers of stacked bricks behind the vehicle (fine and coarse tessellation alternating, two of them hollow and open to one side), joined at
    # the top by a long thin beam; an antenna on one of them
    for ti, (tx, ty) in enumerate(((-0.78, -0.55), (-0.78, 0.55), (0.05, 0.72), (0.62, 0.74))):
        z = -0.53; nb = (7, 6, 6, 4)[ti]
        for k in range(nb):
            hgt = 0.2 if (k + ti) % 3 else 0.12
            fine = (k + ti) % 2 == 0
            hollow = (k == 2 and ti < 2)
            brick((tx, ty, z + hgt / 2), (0.44 if ti < 2 else 0.5, 0.44 if ti < 2 else 0.3, hgt), (36, 36, 16) if fine else (2, 2, 1), stud=(k == nb - 1), skip=(("+x",) if hollow else ()), tubes=hollow)
            z += hgt
        if ti == 1:
            B.cylinder((tx, ty, z + 0.024), 2, 0.01, 0.14, seg(8), r(12), cap1=True)
    B.box((-0.78, 0.0, 0.5), (0.1, 1.3, 0.03), (r(4), r(60), r(2)), bump=None)
    # loose small parts scattered on the plate (seeded): clusters of small bricks at random places / sizes
    for _ in range(40):
        cx, cy = rng.uniform(-0.4, 0.9), rng.choice([-1, 1]) * rng.uniform(0.95, 1.1)
        sz = rng.uniform(0.03, 0.09, 3)
        B.box((cx, cy * 0.78, -0.53 + sz[2] / 2 + 0.26), sz, tuple(r(int(q)) for q in rng.integers(1, 12, 3)), bump=None)


MODEL_SCALE = (0.8, 0.8, 1.3, 0.05)


def make_mesh_clustered(target_tris=300000, seed=0, axis_aligned_plates=3):
    """Lego-like synthetic mesh, ~target_tris triangles (default 3e5 = main.py:133's decimation target), deterministic for (target_tris, seed).
    Returns (verts f32[V,3], tris i32[T,3]) inside [-1,1]^3 (`--bound 1`). See the comment above for what it contains and why."""
    s = float(np.sqrt(max(target_tris, 2000) / 3.0e5))
    for _ in range(6):           # tessellation counts are integers: a few fixed-point steps land within a few per cent of the target
        B = _MeshBuf(); _clustered_parts(B, s, np.random.default_rng(seed))
        T = sum(len(f) for f in B.f)
        if abs(T - target_tris) <= 0.03 * target_tris:
            break
        s *= float(np.sqrt(target_tris / T)) if T > 0.2 * target_tris else 1.3
    v = np.concatenate(B.v, 0); f = np.concatenate(B.f, 0)
    # out of the axes: no edge of the assembly stays axis-parallel (the reference's slab test cannot enter zero-thickness boxes, a-7)
    def rot(axis, deg):
        a = np.deg2rad(deg); c, s_ = np.cos(a), np.sin(a)
        R = np.eye(3); i, j = [(1, 2), (2, 0), (0, 1)][axis]
        R[i, i] = c; R[i, j] = -s_; R[j, i] = s_; R[j, j] = c
        return R
    # (stretched first — taller than wide, studs slightly elliptical — so that from the benchmark camera the model covers more than half the frame)
    v = (v * np.array([MODEL_SCALE[0], MODEL_SCALE[1], MODEL_SCALE[2]])) @ (rot(2, 14.0) @ rot(0, 5.0) @ rot(1, -3.0)).T + np.array([0.0, 0.0, MODEL_SCALE[3]])
    # ... and a few plates that ARE axis-aligned (after the rotation), floating beside the model: every triangle of theirs has a zero-thickness box
    P = _MeshBuf()
    for k in range(axis_aligned_plates):
        P.box((-0.1 + 0.4 * k, -0.6 - 0.03 * k, 0.3 + 0.06 * k), (0.3, 0.2, 0.02), (4, 3, 1))
    if axis_aligned_plates:
        pv = np.concatenate(P.v, 0); pf = np.concatenate(P.f, 0) + len(v)
        v = np.concatenate([v, pv], 0); f = np.concatenate([f, pf], 0)
    assert np.abs(v).max() < 1.0, "clustered mesh leaves the unit bound"
    return np.ascontiguousarray(v.astype(np.float32)), np.ascontiguousarray(f.astype(np.int32))


def mesh_by_name(name, subdiv=7):
    """bench.py / scripts: 'icosphere' (SURVEY 8d's synthetic mesh) or 'clustered' (lego-like); both ~3.3e5 triangles at full size."""
    if name == "clustered":
        return make_mesh_clustered(335872 if subdiv >= 7 else max(2000, 20 * 4 ** subdiv + 512))
    return make_mesh(subdiv, 64 if subdiv >= 6 else 16)
