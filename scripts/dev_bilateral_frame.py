"""Dev (GPU box): cost of the --use_bi_de finish (bilateral, 43x43 taps, 5 buffers) in the bench frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._ops import get_ctx
S = M.scene
v, t = S.make_mesh(7, 64)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, 800, 800, 2)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
ctx = get_ctx(g["fx"], g["fy"])
gb = torch.cat((g["depth"], 0.01 + 0.002 * g["depth"]), dim=-1).contiguous()
def frame(gbd):
    RR.render_fused(ctx, W, None, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], 4, 2, 2, 2.0, 0.1, 0.001, 5, gb_depth=gbd)
for name, gbd in (("EAW", None), ("bilateral", gb)):
    frame(gbd); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): frame(gbd)
    torch.cuda.synchronize(); print("%s finish: 4-spp frame %.2f ms" % (name, (time.perf_counter() - t0) / 3 * 1e3))
