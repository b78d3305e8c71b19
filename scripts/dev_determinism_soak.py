"""Dev (GPU box): run-to-run determinism soak of the production frame — the same 1600 x 1600 frame rendered `frames` times in one process (LBVH rebuilt every time,
the stream schedule free to interleave differently each time) and under several stream / batch configurations in fresh processes; every output buffer of every
frame is hashed and all hashes of a mesh must agree.  A race between the chain and the batched stages, a stale work-queue head or an uninitialised pool slot shows
up here as a hash that differs once in many frames.
    python scripts/dev_determinism_soak.py [spp=32] [frames=40]          (one process; prints the distinct hashes and their counts)"""
import hashlib, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._ops import get_ctx
import bench as B
S = M.scene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mesh = os.environ.get("MIRRES_MESH", "icosphere")
dev = torch.device("cuda", 0)
v, t = S.mesh_by_name(mesh)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
mlp = B.make_field(S, torch, dev)
g = harness.build_gbuffer(W, 800, 800, 2, mlp_mat=mlp)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
ctx = get_ctx(g["fx"], g["fy"])
seen = collections.Counter()
for f in range(frames):
    W.update_mesh(W.vrt, W.v_ind)
    outs, _, _ = RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 777)
    if f % 3 == 1:                       # disturb the schedule: unrelated work on another stream while the next frame is enqueued
        s2 = torch.cuda.Stream()
        with torch.cuda.stream(s2):
            junk = torch.randn((4096, 4096), device="cuda") @ torch.randn((4096, 4096), device="cuda")
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for o in outs:
        h.update(o.contiguous().cpu().numpy().tobytes())
    seen[h.hexdigest()[:16]] += 1
print("mesh %s, %d spp, %d frames, MIRRES_STREAMS=%s MIRRES_PT_BATCH=%s csrc_sha %s: %d distinct hash(es): %s" %
      (mesh, spp, frames, os.environ.get("MIRRES_STREAMS", "default"), os.environ.get("MIRRES_PT_BATCH", "default"), B.csrc_sha(), len(seen), dict(seen)))
