"""Evidence run (GPU box; uses the CPU oracle as the checker, like the parity tests): one LARGE many-sample frame of the bench scene rendered by the HIP path and by the
oracle on the same inputs, compared bit for bit in all six output buffers.  The tests hold 1 spp at 1600 x 1600 and ~36 spp at 320 x 320; this is the
product of the two, where the temporal history, the M cap and the batch schedule have all come into play at full size.
    python scripts/dev_parity_big.py [--res 1600 --spp 32]        (the oracle needs about 2.5 s per Msample on 128 host cores)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR
from mirres_restir_nerf_mesh_amd._ops import get_ctx
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
from oracle import oracle as O

p = argparse.ArgumentParser(); p.add_argument("--res", type=int, default=1600); p.add_argument("--spp", type=int, default=32); p.add_argument("--bounces", type=int, default=2)
p.add_argument("--env", default="256x512", help="environment map HxW (BASELINE configs[3]: an external 1024x2048 map)"); p.add_argument("--albedo_scale", default="", help="x,y,z: use_scale on (relighting, renderer_restir.py:404-408)")
a = p.parse_args()
S = M.scene
v, t = S.mesh_by_name(os.environ.get("MIRRES_MESH", "icosphere"))
info, aabb, _, _ = O.bvh_build(v, t)
fx = fy = a.res; N = fx * fy
eye, rd = S.camera_rays(fy, fx)
r = O.trace(info, aabb, v, t, O.make_rays(np.repeat(eye[None], N, 0), rd), True)
occ = r["hit"].astype(np.float32)
nrm = np.where(occ[:, None] > 0, r["normal"], 0).astype(np.float32)
depth = np.linalg.norm(r["pos"] - eye, axis=1).astype(np.float32)
keep = O.Keep()
params, w0, w1, w2 = S.make_matnet_params(seed=0); mn, mx = S.material_min_max()
mat = O.matnet_struct(keep, params, w0, w1, w2, (-1, -1, -1), (1, 1, 1), mn, mx)
km = O.matnet(mat, r["pos"]); kd = km[:, 0:3].copy(); rm = km[:, 4:6].copy()
eh, ew = (int(x) for x in a.env.split("x")); env = S.make_env(eh, ew)
use_scale = bool(a.albedo_scale); scale = tuple(float(x) for x in a.albedo_scale.split(",")) if use_scale else (1.0, 1.0, 1.0)
if use_scale: kd = (kd * np.array(scale, np.float32)[None, :]).astype(np.float32)      # the primary albedo is scaled by the caller (nerf/renderer.py:1086-1089)
t0 = time.time()
ref = O.render(fx, fy, a.spp, 12345, (info, aabb), v, t, env, occ, nrm, depth, kd, rm, rd, r["pos"], mat=mat, max_bounce=a.bounces, use_scale=use_scale, scale=scale)
t_cpu = time.time() - t0
cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
W = RR.restirbvhWorker(cu(v), cu(t)); W.update_mesh(W.vrt, W.v_ind)
mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(cu(mn), cu(mx)))
with torch.no_grad():
    mlp.encoder.params.copy_(cu(params))
    for i, w in zip((0, 2, 4), (w0, w1, w2)): mlp.net.net[i].weight.copy_(cu(w))
ctx = get_ctx(fx, fy, max_bounce=a.bounces)
torch.cuda.synchronize(); t0 = time.time()
outs, _, _ = RR.render_fused(ctx, W, mlp, use_scale, scale, cu(env), cu(occ[:, None].copy()), cu(nrm), cu(depth[:, None]), cu(kd), cu(rm), cu(rd), cu(r["pos"]), a.spp, 2, 2, 2.0, 0.1, 0.001, 12345)
torch.cuda.synchronize(); t_gpu = time.time() - t0
names = ["final_color", "diffuse", "spec", "indirect", "indirect_diff", "indirect_spec"]
print("%d x %d px, %d spp, %d indirect bounces, env %s%s, hash-grid + MLP material field, T = %d: oracle %.1f s on %d cores, HIP %.3f s (first frame of the context)" % (fx, fy, a.spp, a.bounces, a.env, (", albedo scale " + a.albedo_scale) if use_scale else "", len(t), t_cpu, O.num_threads(), t_gpu))
bad = 0
for o, n in zip(outs, names):
    g = o.cpu().numpy(); e = ref[n]
    diff = g.view(np.uint32) != e.view(np.uint32)
    px = int(diff.any(axis=1).sum()); bad += px
    print("  %-14s pixels with a differing bit: %d of %d   max |diff| %.3e" % (n, px, N, float(np.abs(g - e).max())))
print("BIT-EQUAL" if bad == 0 else "DIFFERENT")
sys.exit(0 if bad == 0 else 1)
