"""Dev (GPU box, ONE rank over RCCL): what the per-sample halo exchange of the exact strip scheme costs on the HOST and on the DEVICE.

dist.render_strips' exchange is issued from mirres_render's host callback once per sample: ctypes -> Python -> torch.distributed.batch_isend_irecv -> return. The
scaling table (scripts/dev_strip_table.py) priced it at an ASSUMED 44 us of device time and never looked at the host. Here strip 4 of 8 of the bench frame is rendered
with the callback doing, per sample:
    none      nothing (the strip's own period)
    copy      the same bytes moved by two torch slice copies on the stream (no RCCL)
    rccl      dist.exchange_halos over RCCL with the peer = this rank (the self exchange of tests/test_gpu_rccl.py: 2 sends + 2 receives of 30 rows x fx x 32 B)
    native    the same sends / receives issued by mirres_render itself (csrc/comm.hip: ncclSend / ncclRecv in one group, the library's own communicator; no callback)
and reports: host microseconds per callback (perf_counter inside the callback), device microseconds between two events around the exchange on the callback's stream
(every 16th sample), and the strip's period per sample with each callback — all against the strip's chain period.

    HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 python scripts/dev_halo_host_cost.py [spp=512]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, dist as D, _lib
from mirres_restir_nerf_mesh_amd._ops import get_ctx
import bench as B
S = M.scene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mesh = os.environ.get("MIRRES_MESH", "icosphere")
dev = torch.device("cuda", 0)
v, t = S.mesh_by_name(mesh)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
mlp = B.make_field(S, torch, dev)
g = harness.build_gbuffer(W, 800, 800, 2, mlp_mat=mlp)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
fx, fy = g["fx"], g["fy"]
world, rank = 8, 4
bounds = D.strip_bounds(fy, world, g["occ"], fx)
y0, y1, lo, hi = D.strip_rows(fy, rank, world, bounds=bounds)
rows = hi - lo; rows_pad = -(-rows // D.STRIP_ROW_QUANTUM) * D.STRIP_ROW_QUANTUM
def _local(x):
    x = x[lo * fx:hi * fx]
    out = torch.zeros((rows_pad * fx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device); out[:rows * fx] = x
    return out
loc = {k: _local(g[k]) for k in ("occ", "normal", "depth", "kd", "rm", "ray_dir", "pos")}
ctx = get_ctx(fx, rows_pad)
# the plan of rank 4 of 8 with both peers replaced by this rank: what is sent upwards comes back as the lower halo and vice versa (same bytes, same calls)
plan = [(0, send, recv) for peer, send, recv in D.halo_plan(fy, fx, rank, world, bounds=bounds)]
nbytes = sum((sb - sa) * fx * 32 for _, (sa, sb), _ in plan)
host_us, dev_pairs = [], []

def make_cb(kind):
    def cb(user, records, sample, stream):
        t0 = time.perf_counter()
        if kind != "none":
            with D.on_stream(stream):
                view = D.device_view(records, (rows_pad, fx, 8))
                ev = None
                if sample % 16 == 0:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)); ev[0].record(torch.cuda.current_stream())
                if kind == "rccl":
                    D.exchange_halos(view, plan)
                else:
                    for _, (sa, sb), (ra, rb) in plan: view[ra:rb].copy_(view[sa:sb])
                if ev: ev[1].record(torch.cuda.current_stream()); dev_pairs.append(ev)
        host_us.append((time.perf_counter() - t0) * 1e6)
        return 0
    return _lib.HALO_FN(cb)

comm = D.native_comm()
def strip(cb):
    W.update_mesh(W.vrt, W.v_ind)
    kw = {"halo_native": (comm, plan, 16)} if cb == "native" else {"halo": cb}
    RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, loc["occ"].clone(), loc["normal"], loc["depth"], loc["kd"], loc["rm"], loc["ray_dir"], loc["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 12345,
                    strip=(fy, lo, y0 - lo, y1 - lo), **kw)

print("mesh %s, strip %d of %d: own rows [%d, %d), local frame %d rows (padded %d) x %d px, %d spp; the exchange moves %.2f MB per sample in %d send + %d receive; csrc_sha %s" %
      (mesh, rank, world, y0, y1, rows, rows_pad, fx, spp, nbytes / 1e6, len(plan), len(plan), B.csrc_sha()))
res = {}
import ctypes as C
for kind in ("none", "copy", "rccl", "native", "none", "copy", "rccl", "native"):
    cb = "native" if kind == "native" else make_cb(kind)
    strip(cb); torch.cuda.synchronize()
    host_us.clear(); dev_pairs.clear()
    t0 = time.perf_counter(); strip(cb); enq = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    h = np.array(host_us) if host_us else np.array([0.0]); d = np.array([a.elapsed_time(b) * 1e3 for a, b in dev_pairs]) if dev_pairs else np.array([0.0])
    if kind == "native":      # no callback: the host cost is inside the frame's enqueue time; the library timed every 16th exchange
        ms, n = C.c_double(0.0), C.c_int(0); _lib.check(_lib.lib().mirres_ctx_halo_time(ctx.h, C.byref(ms), C.byref(n)), "halo_time")
        d = np.array([ms.value * 1e3 / max(1, n.value)] * max(1, n.value))
    print("%-5s period %7.1f us per sample (host enqueue of the frame %7.1f us per sample) | callback on the host: mean %6.1f us, median %6.1f, p95 %6.1f | exchange on the device: mean %6.1f us, median %6.1f (n = %d)" %
          (kind, dt * 1e6 / spp, enq * 1e6 / spp, h.mean(), np.median(h), np.percentile(h, 95), d.mean(), np.median(d), len(d)))
dist.barrier(); dist.destroy_process_group()
