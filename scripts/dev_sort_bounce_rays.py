"""Dev (GPU box): what ORDER is worth to the traversal kernels on the rays of the SECOND indirect vertex.  The rays of the first indirect vertex start at primary
hits (pixel order = coherent origins); those of the second start at the first vertex's hit points, which — in slot order — jump all over the scene.  Here: first-
vertex hit points of the bench view (K directions per foreground pixel), then one closest-hit ray and two shadow-like rays per point, traced by the production
kernels (mirres_bvh_trace mode 2 = ordered closest hit, mode 0 = shadow rays) in slot order, sorted by a 24-bit Morton key of the origin (the order the material
lookup's position sort already produces for these very slots), sorted by 30 bits, and in a random order.  Sort cost not included (see k_ls_* in the kernel trace:
about 0.2 ms per pass and 7.5 M entries).        MIRRES_MESH=clustered python scripts/dev_sort_bounce_rays.py [K=4]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._lib import lib, check
S = M.scene
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mesh = os.environ.get("MIRRES_MESH", "icosphere")
v, t = S.mesh_by_name(mesh)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, 800, 800, 2)
fg = g["occ"][:, 0] > 0.5
pos, nrm = g["pos"][fg], g["normal"][fg]
gen = torch.Generator(device="cuda").manual_seed(0)
def hemi(n_):
    r = torch.nn.functional.normalize(torch.randn(n_.shape, device="cuda", generator=gen), dim=1)
    return torch.nn.functional.normalize(n_ + 0.98 * r, dim=1)
P, Nn = [], []
for k in range(K):                      # sample-major like a batch
    d = hemi(nrm)
    r = W.trace(pos + 0.01 * d, d, closest=True)
    h = r["hit"] > 0
    P.append(r["pos"][h]); Nn.append(torch.nn.functional.normalize(r["normal"][h], dim=1))
P = torch.cat(P).contiguous(); Nn = torch.cat(Nn).contiguous(); n = P.shape[0]
# the normal the engine shades with faces the incoming ray; for ray statistics a random side does as well: directions over the hemisphere of +-normal
D1 = hemi(Nn); O1 = P + 0.001 * D1
def rays_of(o, d):
    r = torch.empty((o.shape[0], 8), device="cuda"); r[:, 0:3] = o; r[:, 3] = 0; r[:, 4:7] = d; r[:, 7] = 1e7; return r
def time_mode(rays, mode, L=5):
    k = rays.shape[0]
    hit = torch.zeros(k, dtype=torch.int32, device="cuda"); tt = torch.zeros(k, device="cuda"); pp = torch.zeros((k, 3), device="cuda"); nn_ = torch.zeros((k, 3), device="cuda"); pr = torch.zeros(k, dtype=torch.int32, device="cuda")
    if mode == 0:
        f = lambda: check(lib().mirres_bvh_trace(W.h, rays.data_ptr(), k, 0, hit.data_ptr(), None, None, None, None, None, None), "t")
    else:
        f = lambda: check(lib().mirres_bvh_trace(W.h, rays.data_ptr(), k, mode, hit.data_ptr(), tt.data_ptr(), pp.data_ptr(), nn_.data_ptr(), pr.data_ptr(), None, None), "t")
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(L): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / L, int(hit.sum())
def morton(p, bits):
    q = ((p * 0.5 + 0.5).clamp(0, 1 - 1e-6) * (1 << bits)).long()
    def ex(x):
        x = (x | (x << 16)) & 0x30000FF; x = (x | (x << 8)) & 0x300F00F; x = (x | (x << 4)) & 0x30C30C3; x = (x | (x << 2)) & 0x9249249; return x
    return ex(q[:, 0]) << 2 | ex(q[:, 1]) << 1 | ex(q[:, 2])
orders = {"slot order (production)": torch.arange(n, device="cuda"), "sorted by 8-bit-per-axis Morton of the origin": torch.argsort(morton(P, 8), stable=True),
          "sorted by 10-bit-per-axis Morton": torch.argsort(morton(P, 10), stable=True), "random permutation": torch.randperm(n, device="cuda", generator=gen)}
print("%s: %d second-vertex rays (%d samples x %d foreground px x hit fraction)" % (mesh, n, K, int(fg.sum())))
for name, perm in orders.items():
    ra = rays_of(O1[perm], D1[perm])
    ms_c, hc = time_mode(ra, 2)
    ms_a, ha = time_mode(ra, 0)
    print("%-48s closest %7.3f ms (%6.2f Grays/s, %d hits)   shadow %7.3f ms (%6.2f Grays/s, %d hits)" % (name, ms_c, n / ms_c / 1e6, hc, ms_a, n / ms_a / 1e6, ha))
