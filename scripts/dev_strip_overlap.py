"""Dev (GPU box): is the per-sample halo exchange worth taking off the chain?  One rank's strip of a `world`-GPU run at 128 spp with a SYNTHETIC exchange (a spin of
`us` microseconds on the stream the callback is given, standing in for the two point-to-point transfers over xGMI) in three set-ups:
  none     no exchange at all (lower bound)
  exposed  exchange in line on the chain, between temporal and spatial reuse (mirres_render_args_t.strip_overlap = 0; the next sample's temporal merge fused into the resolve)
  hidden   exchange on the side stream behind the interior rows' spatial pass, border rows afterwards (strip_overlap = 1: three more launches per sample, no fusion)
    python scripts/dev_strip_overlap.py [world=8] [us=50] [spp=128]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, dist as D, _lib
from mirres_restir_nerf_mesh_amd._ops import get_ctx
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
S = M.scene
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
v, t = S.mesh_by_name(os.environ.get("MIRRES_MESH", "icosphere"))
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, 800, 800, 2)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
fx, fy = g["fx"], g["fy"]
mn, mx = S.material_min_max()
mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=3)
# calibrate torch.cuda._sleep: cycles per microsecond
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000000); torch.cuda.synchronize(); e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 10000000 / (e0.elapsed_time(e1) * 1e3)
spin = int(us * cyc_per_us)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
bounds = D.strip_bounds(fy, world, g["occ"], fx)
rank = world // 2                      # a middle strip: neighbours on both sides
y0, y1, lo, hi = D.strip_rows(fy, rank, world, bounds=bounds)
loc = {k: g[k][lo * fx:hi * fx].contiguous() for k in ("occ", "normal", "depth", "kd", "rm", "ray_dir", "pos")}
ctx = get_ctx(fx, hi - lo)
def cb_none(u, r, s, st): return 0
def cb_spin(u, r, s, st):
    with D.on_stream(st):
        torch.cuda._sleep(spin)
    return 0
def frame(cb, overlap):
    h = _lib.HALO_FN(cb)
    def f():
        W.update_mesh(W.vrt, W.v_ind)
        RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, loc["occ"].clone(), loc["normal"], loc["depth"], loc["kd"], loc["rm"], loc["ray_dir"], loc["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 1,
                        strip=(fy, lo, y0 - lo, y1 - lo), halo=h, strip_overlap=overlap)
    return f
print("strip %d of %d: own rows [%d,%d) (%d), local frame %d rows, %d spp, synthetic exchange %.0f us (%d spin cycles)" % (rank, world, y0, y1, y1 - y0, hi - lo, spp, us, spin))
res = {}
for rep in range(2):
    for name, cb, ov in (("none", cb_none, False), ("exposed", cb_spin, False), ("hidden", cb_spin, True), ("split, no exchange", cb_none, True)):
        res.setdefault(name, []).append(timed(frame(cb, ov)))
for name, ms in res.items():
    m = min(ms)
    print("%-20s %8.2f ms/frame  %7.1f us per sample  (+%.1f us per sample over `none`)" % (name, m, m / spp * 1e3, (m - min(res["none"])) / spp * 1e3))
