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
