"""Dev (GPU box): per-pixel finite differences of FinalShading w.r.t. the normal against the analytic adjoint (the op is pixel-local, so one perturbed
evaluation gives every pixel's directional derivative). Lists the pixels where they disagree: kinks of the shading model or an adjoint error?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
os.environ.setdefault("MIRRES_TEST_SEED", "3")
from oracle import oracle as O
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, Resampling as RS
from util import SmallFrame
F = SmallFrame(O, M.scene, fx=40, fy=32)
W = RR.restirbvhWorker(torch.from_numpy(F.vert).cuda(), torch.from_numpy(F.tri).cuda()); W.update_mesh(W.vrt, W.v_ind)
mods = RR.load_m_for_restir(F.fx, F.fy)
N = F.N
tile_ld, _, tile_pdf = O.light_tiles(F.frame, 300)
r0 = O.new_reservoirs(N); O.initial(F.frame, r0, tile_ld, tile_pdf, 302)
vis = O.final_vis(F.frame, r0); fdir, fdist, fLi = O.eval_final(F.frame, r0, vis)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
wts = [torch.rand((N, 3), device="cuda", generator=g) for _ in range(3)]
def per_pixel(normal):
    c, d, s = RS.FinalShading.apply(mods[6], cu(fdir), cu(fdist[:, None]), cu(fLi), cu(F.tex), F.Wc, F.Hc, F.fx, F.fy, cu(F.occ[:, None]), normal, cu(F.ray_dir), cu(F.kd), cu(F.rm))
    return (c.double() * wts[0]).sum(1) + (d.double() * wts[1]).sum(1) + (s.double() * wts[2]).sum(1)
n0 = cu(F.normal).requires_grad_(True)
per_pixel(n0).sum().backward()
d = torch.randn((N, 3), device="cuda", generator=g) * 0.5
ana = (n0.grad.double() * d).sum(1)
for eps in (2e-3, 5e-4, 1e-4):
    with torch.no_grad():
        num = (per_pixel((n0.detach().double() + eps * d).float()) - per_pixel((n0.detach().double() - eps * d).float())) / (2 * eps)
    err = (ana - num).abs()
    bad = err > 0.05 * num.abs() + 1e-3
    print("eps %.0e: sum ana %.4f num %.4f; pixels off by > 5%%: %d of %d (foreground %d); their share of |ana - num|: %.3f" % (eps, float(ana.sum()), float(num.sum()), int(bad.sum()), N, int((F.occ > 0.5).sum()), float(err[bad].sum() / err.sum())))
idx = torch.nonzero(bad)[:6, 0]
for i in idx.tolist():
    nn = F.normal[i]; wi = -F.ray_dir[i]
    print("pixel %d: ana %.5f num %.5f  n.z %.4f  n.v %.4f  n.l %.4f  rough %.3f" % (i, float(ana[i]), float(num[i]), nn[2], float((nn * wi).sum()), float((nn * fdir[i]).sum()), F.rm[i, 0]))
