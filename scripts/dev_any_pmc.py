"""Dev (GPU box): the production shadow-ray kernel (k_trace_any4q) alone on a frame-like ray set — K rays per foreground pixel of the bench view,
origins 0.01 off the surface, directions over the hemisphere, queue order pixel-major (what the spatial / bounce stages emit).  Prints the
event-timed launch; run under `rocprofv3 --pmc ...` (scripts/pmc_any.sh) for the counters of exactly this launch.

    python scripts/dev_any_pmc.py [res=1600] [K=7] [launches=5]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._lib import lib, check
S = M.scene
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
K = int(sys.argv[2]) if len(sys.argv) > 2 else 7
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
v, t = S.mesh_by_name(os.environ.get("MIRRES_MESH", "icosphere"))
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind); W.upgrade()      # the hierarchy a long frame traverses (round 6: the SAH top is the build's second step)
g = harness.build_gbuffer(W, res, res, 1)
fg = g["occ"][:, 0] > 0.5
pos, nrm = g["pos"][fg], g["normal"][fg]
gen = torch.Generator(device="cuda").manual_seed(0)
n = pos.shape[0]
r = torch.randn((n, K, 3), device="cuda", generator=gen); r = r / r.norm(dim=2, keepdim=True)
d = nrm[:, None, :] + 0.98 * r; d = d / d.norm(dim=2, keepdim=True)
o = pos[:, None, :] + 0.01 * d
k = n * K
rays = torch.empty((k, 8), device="cuda"); rays[:, 0:3] = o.reshape(k, 3); rays[:, 3] = 0; rays[:, 4:7] = d.reshape(k, 3); rays[:, 7] = 1e7
hit = torch.zeros(k, dtype=torch.int32, device="cuda")
tt = torch.zeros(k, device="cuda"); p = torch.zeros((k, 3), device="cuda"); nn = torch.zeros((k, 3), device="cuda"); pr = torch.zeros(k, dtype=torch.int32, device="cuda")
def launch():
    check(lib().mirres_bvh_trace(W.h, rays.data_ptr(), k, mode, hit.data_ptr(), tt.data_ptr(), p.data_ptr(), nn.data_ptr(), pr.data_ptr(), None, None), "t")
launch(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(L):
    launch()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / L
print("mode %d: %d rays, %.3f ms per launch, %.2f Grays/s, hit fraction %.3f, checksum %d" % (mode, k, ms, k / ms / 1e6, float(hit.float().mean()), int(hit.sum())))
