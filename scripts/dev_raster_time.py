"""Dev (GPU box): what raster.dr.rasterize costs at the headline frame's internal size (1600 x 1600 primary rays from the near plane, conventional closest hit =
mirres_bvh_trace mode 4 on the reference-order kernel) next to the ordered fast path (mode 2) on the same rays.
    python scripts/dev_raster_time.py      MIRRES_MESH=clustered python scripts/dev_raster_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, raster, harness
from mirres_restir_nerf_mesh_amd._lib import lib, check
S = M.scene
mesh = os.environ.get("MIRRES_MESH", "icosphere")
v, t = S.mesh_by_name(mesh)
vert = torch.from_numpy(v).cuda(); tri = torch.from_numpy(t).cuda()
W = RR.restirbvhWorker(vert, tri); W.update_mesh(W.vrt, W.v_ind)
H = Wd = 1600
eye = np.array([2.2, -1.6, 1.5]) * 1.2; fwd = -eye / np.linalg.norm(eye)
right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right); up = np.cross(right, fwd)
pose = np.eye(4); pose[:3, 0] = right; pose[:3, 1] = up; pose[:3, 2] = -fwd; pose[:3, 3] = eye
mvp = harness.mvp_from_pose(torch.from_numpy(pose.astype(np.float32)).cuda(), (1800.0, 1800.0, Wd / 2.0, H / 2.0), H, Wd)
pos_clip = (torch.cat((vert, torch.ones_like(vert[:, :1])), 1) @ mvp.t()).unsqueeze(0)
glctx = raster.dr.RasterizeCudaContext(W)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rast = None
def full():
    global rast
    rast, _ = raster.dr.rasterize(glctx, pos_clip, tri, (H, Wd), mvp=mvp)
ms = timed(full)
cov = float((rast[..., 3] > 0).float().mean())
# the same primary rays (from the eye) through the two closest-hit kernels
n = H * Wd
xs = (2 * torch.arange(Wd, device="cuda") + 1) / Wd - 1; ys = (2 * torch.arange(H, device="cuda") + 1) / H - 1
gy, gx = torch.meshgrid(ys, xs, indexing="ij")
Mi = torch.inverse(mvp.double())
pn = torch.stack((gx.double().reshape(-1), gy.double().reshape(-1), torch.full((n,), -1.0, device="cuda", dtype=torch.float64), torch.ones(n, device="cuda", dtype=torch.float64)), 1) @ Mi.t()
pf = torch.stack((gx.double().reshape(-1), gy.double().reshape(-1), torch.full((n,), 0.9, device="cuda", dtype=torch.float64), torch.ones(n, device="cuda", dtype=torch.float64)), 1) @ Mi.t()
pn = pn[:, :3] / pn[:, 3:]; pf = pf[:, :3] / pf[:, 3:]
rays = torch.empty((n, 8), device="cuda"); rays[:, 0:3] = pn.float(); rays[:, 3] = 0; rays[:, 4:7] = (pf - pn).float(); rays[:, 7] = 1e7
hit = torch.zeros(n, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, device="cuda"); pr = torch.zeros(n, dtype=torch.int32, device="cuda")
res = {}
for mode in (2, 4, 1):
    def tr():
        check(lib().mirres_bvh_trace(W.h, rays.data_ptr(), n, mode, hit.data_ptr(), tt.data_ptr(), None, None, pr.data_ptr(), None, None), "trace")
    res[mode] = timed(tr); 
    if mode == 2: p2 = pr.clone(); h2 = hit.clone()
    if mode == 4: same = float(((pr == p2) | (h2 == 0)).float().mean()); sameh = bool(torch.equal(hit, h2))
print("mesh %s, %d x %d: dr.rasterize %.2f ms (coverage %.3f); the primary rays alone: ordered fast path (mode 2) %.2f ms, conventional closest hit (mode 4) %.2f ms, reference order (mode 1) %.2f ms; "
      "modes 2 and 4 agree on the hit bit: %s, on the triangle of %.5f of the hits" % (mesh, Wd, H, ms, cov, res[2], res[4], res[1], sameh, same))
