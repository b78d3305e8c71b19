"""Dev (GPU box): host time to ENQUEUE one frame (the mirres_render call returns when everything is queued) against the GPU time of the frame —
is the frame launch-bound?   python scripts/dev_enqueue_time.py [res=800] [ssaa=2] [spp=128]"""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._ops import get_ctx
res = int(sys.argv[1]) if len(sys.argv) > 1 else 800; ssaa = int(sys.argv[2]) if len(sys.argv) > 2 else 2; spp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
S = M.scene
v, t = S.make_mesh(7, 64)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, res, res, ssaa)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
ctx = get_ctx(g["fx"], g["fy"])
def frame():
    return RR.render_fused(ctx, W, None, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 7)
frame(); torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); frame(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%dx%d internal, %d spp: host enqueue %.1f ms, frame %.1f ms (enqueue = %.1f %% of the frame)" % (g["fx"], g["fy"], spp, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 100 * (t1 - t0) / (t2 - t0)))
