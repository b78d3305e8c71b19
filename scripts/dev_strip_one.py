"""Dev (GPU box): ONE strip of the exact row-strip scheme rendered alone, for a kernel trace of where a strip's time goes.
    python scripts/dev_strip_one.py world rank spp [frames=3] [bg]      (world=1: the whole frame; `bg`: occupancy zeroed = the pure fixed cost)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, dist as D, _lib
from mirres_restir_nerf_mesh_amd._ops import get_ctx
import bench as B
S = M.scene
world, rank, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3
bg = len(sys.argv) > 5 and sys.argv[5] == "bg"
dev = torch.device("cuda", 0)
v, t = S.mesh_by_name(os.environ.get("MIRRES_MESH", "icosphere"))
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
mlp = B.make_field(S, torch, dev)
g = harness.build_gbuffer(W, 800, 800, 2, mlp_mat=mlp)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
fx, fy = g["fx"], g["fy"]
if bg:
    g["occ"] = torch.zeros_like(g["occ"])
if world == 1:
    ctx = get_ctx(fx, fy)
    def frame():
        W.update_mesh(W.vrt, W.v_ind)
        RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 12345)
else:
    bounds = D.strip_bounds(fy, world, None if bg else g["occ"], fx)
    y0, y1, lo, hi = D.strip_rows(fy, rank, world, bounds=bounds)
    rows = hi - lo; rows_pad = -(-rows // D.STRIP_ROW_QUANTUM) * D.STRIP_ROW_QUANTUM
    def _local(x):
        x = x[lo * fx:hi * fx]
        out = torch.zeros((rows_pad * fx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device); out[:rows * fx] = x
        return out
    loc = {k: _local(g[k]) for k in ("occ", "normal", "depth", "kd", "rm", "ray_dir", "pos")}
    ctx = get_ctx(fx, rows_pad)
    cb = _lib.HALO_FN(lambda u, r, s, st: 0)
    def frame():
        W.update_mesh(W.vrt, W.v_ind)
        RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, loc["occ"].clone(), loc["normal"], loc["depth"], loc["kd"], loc["rm"], loc["ray_dir"], loc["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 12345,
                        strip=(fy, lo, y0 - lo, y1 - lo), halo=cb)
frame(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    frame()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / frames
print("world %d rank %d%s: %d spp, %.2f ms per frame, %.1f us per sample" % (world, rank, " (background only)" if bg else "", spp, dt * 1e3, dt * 1e6 / spp))
