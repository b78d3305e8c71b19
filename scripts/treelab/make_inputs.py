"""Inputs for scripts/treelab/treelab.cpp (CPU, uses the oracle for the G-buffer and the reference LBVH): mesh.bin, rays.bin (shadow-like rays from the
bench view: per foreground pixel K rays, half towards environment-importance-sampled directions, half over the hemisphere), lbvh.bin.
    python scripts/treelab/make_inputs.py <icosphere|clustered> <outdir> [res=400] [K=6]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from oracle import oracle as O
S = M.scene
name, out = sys.argv[1], sys.argv[2]
res = int(sys.argv[3]) if len(sys.argv) > 3 else 400
K = int(sys.argv[4]) if len(sys.argv) > 4 else 6
os.makedirs(out, exist_ok=True)
v, t = S.mesh_by_name(name)
info, aabb, _, _ = O.bvh_build(v, t)
with open(os.path.join(out, "mesh.bin"), "wb") as f:
    np.array([len(v), len(t)], np.int32).tofile(f); v.astype(np.float32).tofile(f); t.astype(np.int32).tofile(f)
info.astype(np.int32).tofile(os.path.join(out, "lbvh.bin"))
eye, rd = S.camera_rays(res, res)
r = O.trace(info, aabb, v, t, O.make_rays(np.repeat(eye[None], res * res, 0), rd), True)
fg = r["hit"] > 0
pos, nrm = r["pos"][fg], r["normal"][fg]
n = len(pos)
rng = np.random.default_rng(0)
env = S.make_env(256, 512)
H, Wd = env.shape[:2]
lum = env @ np.array([0.2126, 0.7152, 0.0722])
th = np.pi * (np.arange(H) + 0.5) / H
w = (lum * np.sin(th)[:, None]).reshape(-1); w /= w.sum()
tex = rng.choice(H * Wd, size=n * (K // 2), p=w)
ty, tx = tex // Wd, tex % Wd
theta = np.pi * (ty + rng.random(len(tex))) / H; phi = 2 * np.pi * (tx + rng.random(len(tex))) / Wd
de = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], 1)      # env space (y up)
dw = np.stack([-de[:, 0], de[:, 2], de[:, 1]], 1).reshape(n, K // 2, 3)                        # world: ngp_dir is its own inverse up to the swap
g = rng.normal(size=(n, K - K // 2, 3)); g /= np.linalg.norm(g, axis=2, keepdims=True)
dh = nrm[:, None, :] + 0.98 * g; dh /= np.linalg.norm(dh, axis=2, keepdims=True)
d = np.concatenate([dw, dh], 1)
d = np.where((d * nrm[:, None, :]).sum(2, keepdims=True) < 0, d - 2 * (d * nrm[:, None, :]).sum(2, keepdims=True) * nrm[:, None, :], d)   # into the upper hemisphere (a light behind the surface has target 0)
o = pos[:, None, :] + 0.01 * d
rays = np.zeros((n * K, 8), np.float32); rays[:, 0:3] = o.reshape(-1, 3); rays[:, 4:7] = d.reshape(-1, 3); rays[:, 7] = 1e7
with open(os.path.join(out, "rays.bin"), "wb") as f:
    np.array([len(rays)], np.int32).tofile(f); rays.tofile(f)
print(name, "T", len(t), "foreground px", n, "rays", len(rays))
