"""Dev (GPU box): what ray ORDER is worth to the shadow-ray kernel (north-star: "sorted ray compaction").  Frame-like shadow rays — K per foreground pixel of the
bench view, origins 0.01 off the surface, directions drawn from the environment's importance distribution (what reservoir samples point at: mostly the sun lobe) or
uniformly over the hemisphere (what BSDF samples look like) — traced by the production kernel (mirres_bvh_trace mode 0) in different queue orders:
  pixel     the engine's order today (pixel-major, the K rays of a pixel adjacent)
  oct+tile  key = direction octant | 16x16-pixel tile | pixel   (binning at append time could produce this)
  oct+morton key = direction octant | 30-bit Morton code of the origin
  dir16+morton key = 4x4 octahedral direction cell | Morton code of the origin
  random    a random permutation (the incoherent bound)
Results are indexed by slot, so every order returns the same answers (checksum printed).  Also prints what a device-wide radix sort of that many 64-bit
(key, slot) pairs costs (torch.sort, a proxy for rocPRIM's)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._lib import lib, check
S = M.scene
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
v, t = S.make_mesh(7, 64)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, res, res, 1)
fx = g["fx"]
fgi = torch.nonzero(g["occ"][:, 0] > 0.5)[:, 0]
pos, nrm = g["pos"][fgi], g["normal"][fgi]
n = pos.shape[0]
gen = torch.Generator(device="cuda").manual_seed(0)


def env_dirs(count):
    env = torch.from_numpy(S.make_env(256, 512)).cuda()
    Hc, Wc = env.shape[:2]
    lum = env[..., 0] * 0.212671 + env[..., 1] * 0.715160 + env[..., 2] * 0.072169
    th = (torch.arange(Hc, device="cuda") + 0.5) / Hc * np.pi
    w = (lum * torch.sin(th)[:, None]).flatten()
    idx = torch.multinomial(w / w.sum(), count, replacement=True, generator=gen)
    r, c = idx // Wc, idx % Wc
    u = torch.rand((count, 2), device="cuda", generator=gen)
    theta = (r + u[:, 0]) / Hc * np.pi; phi = (c + u[:, 1]) / Wc * 2 * np.pi
    d_env = torch.stack([torch.sin(theta) * torch.cos(phi), torch.cos(theta), torch.sin(theta) * torch.sin(phi)], 1)     # y-up lat-long direction
    return torch.stack([-d_env[:, 0], d_env[:, 2], d_env[:, 1]], 1)           # inverse of ngp_dir (its own inverse): world z-up


def make(kind):
    if kind == "env":
        d = env_dirs(n * K).reshape(n, K, 3)
    else:
        r = torch.randn((n, K, 3), device="cuda", generator=gen); r = r / r.norm(dim=2, keepdim=True)
        d = nrm[:, None, :] + 0.98 * r; d = d / d.norm(dim=2, keepdim=True)
    keep = (d * nrm[:, None, :]).sum(2) > 1e-3                                 # below the horizon: target function zero, never traced
    o = pos[:, None, :] + 0.01 * d
    pix = fgi[:, None].expand(n, K)
    return o[keep], d[keep], pix[keep]


def morton3(o):
    lo = o.min(0).values; hi = o.max(0).values
    q = ((o - lo) / (hi - lo) * 1023).clamp(0, 1023).long()
    def ex(x):
        x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
        return x
    return (ex(q[:, 0]) << 2) | (ex(q[:, 1]) << 1) | ex(q[:, 2])


def oct_cell(d, nb):
    l1 = d.abs().sum(1, keepdim=True); p = d / l1
    xy = torch.where(p[:, 2:3] >= 0, p[:, :2], (1 - p[:, [1, 0]].abs()) * torch.where(p[:, :2] >= 0, 1.0, -1.0))
    c = ((xy * 0.5 + 0.5) * nb).clamp(0, nb - 1).long()
    return c[:, 1] * nb + c[:, 0]


def trace_time(o, d, L=5):
    k = o.shape[0]
    rays = torch.empty((k, 8), device="cuda"); rays[:, 0:3] = o; rays[:, 3] = 0; rays[:, 4:7] = d; rays[:, 7] = 1e7
    hit = torch.zeros(k, dtype=torch.int32, device="cuda")
    f = lambda: check(lib().mirres_bvh_trace(W.h, rays.data_ptr(), k, 0, hit.data_ptr(), None, None, None, None, None, None), "t")
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(L):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / L, int(hit.sum())


for kind in ("env", "hemisphere"):
    o, d, pix = make(kind)
    k = o.shape[0]
    octant = ((d[:, 0] < 0).long() << 2) | ((d[:, 1] < 0).long() << 1) | (d[:, 2] < 0).long()
    tile = ((pix // fx) // 16) * ((fx + 15) // 16) + (pix % fx) // 16
    mo = morton3(o)
    orders = {"pixel": None,
              "oct+tile": (octant << 40) | (tile << 20) | torch.arange(k, device="cuda") % (1 << 20),
              "oct+morton": (octant << 30) | mo,
              "dir16+morton": (oct_cell(d, 4) << 30) | mo,
              "random": torch.randperm(k, device="cuda", generator=gen)}
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); torch.sort(orders["oct+morton"]); e1.record(); torch.cuda.synchronize()
    print("%s directions: %d rays (octant histogram %s); torch.sort of the keys %.3f ms" % (kind, k, torch.bincount(octant, minlength=8).tolist(), e0.elapsed_time(e1)))
    base = None
    for name, key in orders.items():
        if key is None:
            ms, cs = trace_time(o, d)
        else:
            perm = torch.argsort(key, stable=True)
            ms, cs = trace_time(o[perm].contiguous(), d[perm].contiguous())
        base = base or ms
        print("   %-13s %.3f ms  %.2f Grays/s  (%+.1f %% vs pixel order)  hits %d" % (name, ms, k / ms / 1e6, 100 * (base / ms - 1), cs))
