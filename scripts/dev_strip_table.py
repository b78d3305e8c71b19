"""Dev (GPU box): the exact row-strip scheme's scaling as far as ONE GPU can tell — every strip r of N in {2, 4, 8} rendered alone (the local frame
dist.render_strips would hand to rank r: own rows + halo rows + padding, no exchange), against the whole frame on the same device.

    T_r             per-sample time of strip r (spp samples, LBVH rebuild included, best of `reps`)
    balance         max_r T_r / mean_r T_r
    predicted       T_full / (max_r T_r + exchange) with exchange = EXCH_US per sample exposed on the chain (measured: profiles/r06_halo_host_cost.txt)

Also prints each strip's pixel counts and fits T_r = a + b * foreground px + c * background px over all strips of the mesh (a = per-sample fixed cost of a strip,
c / b = what dist.strip_bounds calls bg_weight).

With iters > 1 the boundaries are then moved by dist.StripBalancer from the measured T_r (what the ranks of a real run exchange after every frame), `iters` times,
and the last round is the table's second line per N.

    python scripts/dev_strip_table.py [spp=64] [reps=2] [bg_weight=default] [worlds=2,4,8] [iters=1]
    MIRRES_MESH=clustered python scripts/dev_strip_table.py ..."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, dist as D, _lib
from mirres_restir_nerf_mesh_amd._ops import get_ctx
sys.path.insert(0, ROOT)
import bench as B
S = M.scene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bgw = None if len(sys.argv) <= 3 or sys.argv[3] == "default" else float(sys.argv[3])
worlds = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "2,4,8").split(",")]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 1
mesh = os.environ.get("MIRRES_MESH", "icosphere")
# what the per-sample exchange adds to a strip's period, MEASURED over RCCL on one rank with the library's own send / receive group (scripts/dev_halo_host_cost.py,
# profiles/r06_halo_host_cost.txt: strip 4 of 8, 3.07 MB per sample): +43 us (icosphere) / +36 us (lego-like); the Python callback of rounds 1-5 added +95 / +84 us.
# (rounds 4-5 assumed 44 us.)  MIRRES_EXCH_US overrides.
EXCH_US = float(os.environ.get("MIRRES_EXCH_US", "43.0" if mesh == "icosphere" else "36.0"))
dev = torch.device("cuda", 0)
v, t = S.mesh_by_name(mesh)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
mlp = B.make_field(S, torch, dev)
g = harness.build_gbuffer(W, 800, 800, 2, mlp_mat=mlp)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
fx, fy = g["fx"], g["fy"]


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e6 / spp            # us per sample


ctxf = get_ctx(fx, fy)
def full():
    W.update_mesh(W.vrt, W.v_ind)
    RR.render_fused(ctxf, W, mlp, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 12345)
T_full = timed(full)
print("mesh %s, %dx%d, %d spp, csrc_sha %s: whole frame %.1f us per sample (%.1f Msamples/s)" % (mesh, fx, fy, spp, B.csrc_sha(), T_full, fx * fy / T_full))
del ctxf
occ2 = (g["occ"].reshape(fy, fx) > 0.5)
rows_fit, table = [], {}
kw = {} if bgw is None else {"bg_weight": bgw}
for world in worlds:
  bal = D.StripBalancer(fy, world, **kw)
  for it in range(iters):
    bounds = bal.bounds(fx, g["occ"])
    Ts = []
    for rank in range(world):
        y0, y1, lo, hi = D.strip_rows(fy, rank, world, bounds=bounds)
        rows = hi - lo
        rows_pad = -(-rows // D.STRIP_ROW_QUANTUM) * D.STRIP_ROW_QUANTUM
        def _local(x):
            x = x[lo * fx:hi * fx]
            out = torch.zeros((rows_pad * fx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            out[:rows * fx] = x
            return out
        loc = {k: _local(g[k]) for k in ("occ", "normal", "depth", "kd", "rm", "ray_dir", "pos")}
        ctx = get_ctx(fx, rows_pad)
        cb = _lib.HALO_FN(lambda u, r, s, st: 0)
        def strip():
            W.update_mesh(W.vrt, W.v_ind)
            RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, loc["occ"].clone(), loc["normal"], loc["depth"], loc["kd"], loc["rm"], loc["ray_dir"], loc["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 12345,
                            strip=(fy, lo, y0 - lo, y1 - lo), halo=cb)
        T = timed(strip)
        fgp = int(occ2[y0:y1].sum().item()); bgp = (y1 - y0) * fx - fgp
        Ts.append(T)
        if it == 0:
            rows_fit.append((1.0, fgp, bgp, T))
        if it in (0, iters - 1):
            print("  N=%d round %d strip %d: own rows [%4d,%4d) = %4d, local %4d (padded %4d), fg px %7d, bg px %7d: %8.1f us per sample" % (world, it, rank, y0, y1, y1 - y0, rows, rows_pad, fgp, bgp, T))
        del ctx, loc
    bal.update(Ts)
    mx, mean = max(Ts), sum(Ts) / len(Ts)
    pred = T_full / (mx + EXCH_US)
    table["%d/%s" % (world, "static model" if it == 0 else "after %d measured rounds" % it)] = {
                    "bounds": [int(b) for b in bounds], "T_us": [round(x, 1) for x in Ts], "max_over_mean": round(mx / mean, 3), "sum_over_full": round(sum(Ts) / T_full, 3),
                    "predicted_speedup": round(pred, 2), "predicted_speedup_no_exchange": round(T_full / mx, 2), "ideal_balanced_speedup": round(T_full / (mean + EXCH_US), 2)}
    print("N=%d round %d: max %.1f, mean %.1f (max/mean %.3f), sum/full %.3f -> predicted speed-up %.2fx (%.2fx without exchange; %.2fx if perfectly balanced)" %
          (world, it, mx, mean, mx / mean, sum(Ts) / T_full, pred, T_full / mx, T_full / (mean + EXCH_US)))
A = np.array([r[:3] for r in rows_fit], dtype=np.float64); y = np.array([r[3] for r in rows_fit])
coef, res, _, _ = np.linalg.lstsq(A, y, rcond=None)
fit = A @ coef
print("fit T_r = a + b fg + c bg: a = %.1f us per sample, b = %.3f ns per foreground px, c = %.3f ns per background px, c/b = %.3f; worst residual %.1f %%" %
      (coef[0], coef[1] * 1e3, coef[2] * 1e3, coef[2] / coef[1], float(np.max(np.abs(fit - y) / y)) * 100))
print(json.dumps({"mesh": mesh, "spp": spp, "csrc_sha": B.csrc_sha(), "bg_weight": bgw, "T_full_us": round(T_full, 1), "exchange_us": EXCH_US, "table": table,
                  "fit": {"a_us": round(float(coef[0]), 1), "b_ns_per_fg_px": round(float(coef[1] * 1e3), 4), "c_ns_per_bg_px": round(float(coef[2] * 1e3), 4)}}))
