"""Dev (GPU box, library built with -DMR_EXP_PHASES, MIRRES_LIB=ab/libmirres_PHASES.so): cycles a shadow-ray wave spends per phase.
The instrumentation (MR_EXP_PHASES in bvh_trace.hip) left the tree at the end of round 5; check out commit 8981516 to rebuild it."""
import ctypes as C, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._lib import lib
S = M.scene
K = int(sys.argv[1]) if len(sys.argv) > 1 else 7
v, t = S.make_mesh(7, 64)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, 1600, 1600, 1)
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
L = lib()
L.mirres_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; L.mirres_debug_wave_times.restype = C.c_int
def run():
    return L.mirres_bvh_trace(W.h, rays.data_ptr(), k, 0, hit.data_ptr(), None, None, None, None, None, None)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("rays", k, "launch ms", e0.elapsed_time(e1))
L.mirres_debug_wave_times(W.h, None, 1)
run(); torch.cuda.synchronize()
buf = np.zeros(2 * 16384, dtype=np.uint64)
L.mirres_debug_wave_times(W.h, buf.ctypes.data, 0)
ph = buf[:5 * 6144].reshape(-1, 5).astype(np.float64)
ph = ph[ph[:, 4] > 0]
tot, refill, mem, cmp_, it = ph.T
print("waves", len(ph), "mean cycles per wave: total %.0f  refill %.0f (%.1f%%)  fetch+wait %.0f (%.1f%%)  compute %.0f (%.1f%%)  other %.0f (%.1f%%)" % (
    tot.mean(), refill.mean(), 100 * refill.sum() / tot.sum(), mem.mean(), 100 * mem.sum() / tot.sum(), cmp_.mean(), 100 * cmp_.sum() / tot.sum(),
    (tot - refill - mem - cmp_).mean(), 100 * (tot - refill - mem - cmp_).sum() / tot.sum()))
print("iterations per wave %.0f; per iteration: fetch+wait %.0f cycles, compute %.0f cycles, all %.0f" % (it.mean(), mem.sum() / it.sum(), cmp_.sum() / it.sum(), tot.sum() / it.sum()))
print("ray-iterations per wave-iteration (busy lanes): %.1f of 64" % (k * 11.4 / it.sum()))
