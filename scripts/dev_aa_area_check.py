"""Dev (GPU box): the antialiased coverage of a sphere against closed forms — translating the mesh must not change the covered area (autograd derivative ~ 0,
area continuous in the shift), scaling it by (1 + s) changes the area by 2 s A (autograd derivative = 2 A)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, raster
from train_stage1 import orbit_pose
v, t = M.scene.make_mesh(4, 0) if False else M.scene.make_mesh(4, 8)
# the sphere only (drop the ground grid: its triangles come last)
nv_s = 10 * 4 ** 4 + 2; nt_s = 20 * 4 ** 4
v, t = v[:nv_s], t[:nt_s]
verts = torch.from_numpy(v).cuda(); tris = torch.from_numpy(t).cuda()
centre = verts.mean(0)
H = Wd = 96
pose = torch.from_numpy(orbit_pose(30, 30)).cuda(); focal = 0.5 * Wd / np.tan(0.5 * 0.6911); intr = (focal, focal, Wd * 0.5, H * 0.5)
ro, rd = harness.get_rays(pose, intr, H, Wd)
mvp = harness.mvp_from_pose(pose, intr, H, Wd)
topo = raster.antialias_topology(tris)
Wk = RR.restirbvhWorker(verts, tris)
def area(vv, aa=True):
    Wk.update_mesh(vv.detach().contiguous(), tris)
    rast = raster.rasterize_raycast(Wk, ro, rd)
    mask = (rast[:, 3:4] > 0).float()
    if not aa: return mask.sum()
    clip = torch.cat((vv, torch.ones_like(vv[:, :1])), 1) @ mvp.t()
    return raster.antialias(mask.view(1, H, Wd, 1), rast.view(1, H, Wd, 4), clip[None], tris, topology_hash=topo).sum()
right = pose[:3, 0]
print("shift along the image x axis (pixels)   raw area   antialiased area   d(AA area)/d(shift) by autograd")
px = 2 * np.tan(0.5 * 0.6911) * 3.2 / Wd
for k in range(0, 11):
    s = torch.tensor(k * 0.1 * px, device="cuda", requires_grad=True)
    vv = verts + s * right
    a = area(vv); (g,) = torch.autograd.grad(a, s)
    print("  %.1f   %8.1f   %10.3f   %9.4f per unit = %.4f per pixel" % (k * 0.1, float(area(vv, False)), float(a), float(g), float(g) * px))
A0 = float(area(verts))
s = torch.tensor(0.0, device="cuda", requires_grad=True)
a = area(centre + (verts - centre) * (1 + s)); (g,) = torch.autograd.grad(a, s)
print("scale: area %.2f, autograd d(area)/ds = %.2f, closed form 2 A = %.2f" % (float(a), float(g), 2 * A0))
for ds in (0.01, 0.02, 0.05):
    ap = float(area(centre + (verts - centre) * (1 + ds))); am = float(area(centre + (verts - centre) * (1 - ds)))
    print("   finite difference ds=%.2f: %.2f" % (ds, (ap - am) / (2 * ds)))
