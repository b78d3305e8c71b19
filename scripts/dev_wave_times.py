"""Dev (GPU box): per-wave start/end wall-clock times of one shadow-ray launch (k_trace_any4q), to see ramp-up / tail / imbalance."""
import ctypes as C, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._lib import lib
S = M.scene
v, t = S.make_mesh(7, 64)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, 1600, 1600, 1)
fg = g["occ"][:, 0] > 0.5
pos, nrm = g["pos"][fg], g["normal"][fg]
gen = torch.Generator(device="cuda").manual_seed(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # rays per foreground pixel (7 = the spatial pass's launch size)
pos = pos.repeat_interleave(K, 0); nrm = nrm.repeat_interleave(K, 0)
n = pos.shape[0]
r = torch.randn((n, 3), device="cuda", generator=gen); r = r / r.norm(dim=1, keepdim=True)
d = nrm + 0.98 * r; d = d / d.norm(dim=1, keepdim=True)
o = pos + 0.01 * d
rays = torch.empty((n, 8), device="cuda"); rays[:, 0:3] = o; rays[:, 3] = 0; rays[:, 4:7] = d; rays[:, 7] = 1e7
hit = torch.zeros(n, dtype=torch.int32, device="cuda")
L = lib()
L.mirres_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; L.mirres_debug_wave_times.restype = C.c_int
def run():
    return L.mirres_bvh_trace(W.h, rays.data_ptr(), n, 0, hit.data_ptr(), None, None, None, None, None, None)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("rays", n, "launch ms", e0.elapsed_time(e1))
L.mirres_debug_wave_times(W.h, None, 1)
run(); torch.cuda.synchronize()
buf = np.zeros(2 * 16384, dtype=np.uint64)
L.mirres_debug_wave_times(W.h, buf.ctypes.data, 0)
tt = buf.reshape(-1, 2).astype(np.int64)
tt = tt[tt[:, 0] > 0]
t0 = tt[:, 0].min()
st = (tt[:, 0] - t0) / 100.0; en = (tt[:, 1] - t0) / 100.0      # wall_clock64: 100 MHz -> us
print("waves", len(tt), "span us", en.max())
for q in (0, 1, 5, 25, 50, 75, 95, 99, 100):
    print(f"  pct {q:3d}: start {np.percentile(st, q):8.1f} us   end {np.percentile(en, q):8.1f} us   dur {np.percentile(en - st, q):8.1f}")
print("mean alive fraction", float((en - st).sum() / (len(tt) * en.max())))
edges = np.linspace(0, en.max(), 21)
alive = [(int(((st <= a) & (en > a)).sum())) for a in edges]
print("waves alive at 5 % steps of the launch:", alive)
print("waves started after 50 us:", int((st > 50).sum()), " of them lasting < 20 us:", int(((st > 50) & (en - st < 20)).sum()))
