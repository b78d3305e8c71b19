"""Dev (GPU box): dump_render (BASELINE configs[0] kernel path) at 800 x 800 with a 16 x 32 light probe: ms and (point, light) pairs per second."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, render_dump as RD
from mirres_restir_nerf_mesh_amd.render_helper import generate_envir_map_dir
S = M.scene
v, t = S.make_mesh(7, 64)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, 800, 800, 1)
m = g["occ"][:, 0] > 0.5
pos, nrm, rd, kd = g["pos"][m], g["normal"][m], g["ray_dir"][m], g["kd"][m]
n = pos.shape[0]
rough = torch.full((n, 3), 0.5, device="cuda"); fres = torch.full((n, 3), 0.04, device="cuda")
eh, ew = 16, 32
env = torch.from_numpy(S.make_env(eh, ew)).cuda()
lw, ld = generate_envir_map_dir(eh, ew)
model = types.SimpleNamespace(light_area_weight=lw, fixed_viewdirs=ld)
f = lambda: RD.dump_render(W, pos, nrm, kd, rough, fres, rd, env, eh, ew, model)
out = f(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): out = f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print("dump_render: %d points x %d lights: %.1f ms  (%.2f G pairs/s)  mean rgb %.4f" % (n, ld.shape[0], dt * 1e3, n * ld.shape[0] / dt / 1e9, float(out[0].mean())))
