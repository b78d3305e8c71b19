"""Dev (GPU box): is there slack a two-wide chain could use?  Two whole frames rendered CONCURRENTLY (two contexts, two caller streams: two sample-by-sample chains and
two bulk streams in flight) against the same two frames one after the other.  If the concurrent pair is much faster than twice one frame, the single chain leaves
capacity idle that overlapping consecutive samples (a band-pipelined chain) could reach; if not, the device is saturated by one frame's streams.
    python scripts/dev_two_frames.py [spp=128] [res=800]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, _ops
import bench as B
S = M.scene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
res = int(sys.argv[2]) if len(sys.argv) > 2 else 800
dev = torch.device("cuda", 0)
v, t = S.mesh_by_name(os.environ.get("MIRRES_MESH", "icosphere"))
Ws = []
for i in range(2):      # two BVH handles: every handle has its own traversal work heads
    W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind); Ws.append(W)
mlp = B.make_field(S, torch, dev)
g = harness.build_gbuffer(Ws[0], res, res, 2, mlp_mat=mlp)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
fx, fy = g["fx"], g["fy"]
ctxs = [_ops.Context(fx, fy) for _ in range(2)]
for c in ctxs: c.reserve()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def frame(i):
    RR.render_fused(ctxs[i], Ws[i], mlp, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 12345 + i)
def serial():
    for i in range(2):
        with torch.cuda.stream(streams[0]): frame(i)
def concurrent():
    for i in range(2):
        with torch.cuda.stream(streams[i]): frame(i)
for fn in (serial, concurrent): fn(); torch.cuda.synchronize()
for name, fn in (("one after the other", serial), ("concurrently", concurrent), ("one after the other", serial), ("concurrently", concurrent)):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("two %dx%d x %d spp frames %-20s %8.1f ms  = %7.1f Msamples/s" % (fx, fy, spp, name + ":", dt * 1e3, 2.0 * fx * fy * spp / dt / 1e6))
