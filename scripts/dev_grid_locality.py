"""Dev (GPU box): what would ordering the live-slot list by position buy the fused hash-grid gather + MLP kernel (k_mlp_mfma<1, 2>)?  The indirect vertices of a batch
reach the kernel in slot order (sample-major, pixel-minor); consecutive slots hold hit points of scattered bounce rays. Here: first-bounce hit points of the bench view
(K cosine-ish directions per foreground pixel) through the PRODUCTION kernel in three orders — slot order, sorted by a 30-bit Morton code of the position (the
upper bound of any bucketing scheme; the sort itself is not timed), and a random permutation (the lower bound).   python scripts/dev_grid_locality.py [K=4]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._lib import lib, check
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
S = M.scene
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
v, t = S.mesh_by_name(os.environ.get("MIRRES_MESH", "icosphere"))
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, 800, 800, 2)
fg = g["occ"][:, 0] > 0.5
pos, nrm = g["pos"][fg], g["normal"][fg]
gen = torch.Generator(device="cuda").manual_seed(0)
pts = []
for k in range(K):       # sample-major like a batch: all pixels of sample 0, then of sample 1, ...
    d = torch.nn.functional.normalize(nrm + 0.98 * torch.nn.functional.normalize(torch.randn(nrm.shape, device="cuda", generator=gen), dim=1), dim=1)
    r = W.trace(pos + 0.01 * d, d, closest=True)
    pts.append(r["pos"][r["hit"] > 0])
P = torch.cat(pts).contiguous(); n = P.shape[0]
params, w0, w1, w2 = S.make_matnet_params(seed=0); mn, mx = S.material_min_max()
mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()))
with torch.no_grad():
    mlp.encoder.params.copy_(torch.from_numpy(params).cuda())
    for i, w in zip((0, 2, 4), (w0, w1, w2)): mlp.net.net[i].weight.copy_(torch.from_numpy(w).cuda())
L = lib(); L.mirres_debug_matnet_scatter_mfma.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5; L.mirres_debug_matnet_scatter_mfma.restype = C.c_int
st = mlp._struct()
occ = torch.ones(n, device="cuda"); kd = torch.zeros((n, 3), device="cuda"); rm = torch.zeros((n, 2), device="cuda"); idx = torch.zeros(n, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
def run(p):
    check(L.mirres_debug_matnet_scatter_mfma(C.byref(st), occ.data_ptr(), p.data_ptr(), n, kd.data_ptr(), rm.data_ptr(), idx.data_ptr(), cnt.data_ptr(), None), "scatter")
def timeit(p, reps=10):
    run(p); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run(p)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
def morton(p, bits=10):
    q = ((p * 0.5 + 0.5).clamp(0, 1 - 1e-6) * (1 << bits)).long()
    def ex(x):
        x = (x | (x << 16)) & 0x30000FF; x = (x | (x << 8)) & 0x300F00F; x = (x | (x << 4)) & 0x30C30C3; x = (x | (x << 2)) & 0x9249249; return x
    return ex(q[:, 0]) << 2 | ex(q[:, 1]) << 1 | ex(q[:, 2])
orders = {"slot order (production)": P, "random permutation": P[torch.randperm(n, device="cuda", generator=gen)].contiguous()}
for b in (4, 6, 10):
    orders["sorted by %d-bit-per-axis Morton" % b] = P[torch.argsort(morton(P, b), stable=True)].contiguous()
ref = None
print("%d indirect vertices (%d samples x %d foreground px x hit fraction)" % (n, K, int(fg.sum())))
for name, p in orders.items():
    ms = timeit(p)
    print("%-36s %7.3f ms  %6.1f Mpoints/s" % (name, ms, n / ms / 1e3))
