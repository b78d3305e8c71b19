"""Dev (GPU box): SHA-256 of every output buffer of one frame of the bench scene (1600 x 1600 internal, hash-grid material) — run once per library
build (MIRRES_LIB=...) to check that two builds render the same bits.   python scripts/dev_frame_hash.py [spp=8]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._ops import get_ctx
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
S = M.scene
v, t = S.mesh_by_name(os.environ.get("MIRRES_MESH", "icosphere"))
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
mn, mx = S.material_min_max()
params, w0, w1, w2 = S.make_matnet_params(seed=0)        # numpy-seeded: the same field in every process
mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()))
with torch.no_grad():
    mlp.encoder.params.copy_(torch.from_numpy(params).cuda())
    for i, w in zip((0, 2, 4), (w0, w1, w2)): mlp.net.net[i].weight.copy_(torch.from_numpy(w).cuda())
g = harness.build_gbuffer(W, 800, 800, 2, mlp_mat=mlp)
env = torch.from_numpy(S.make_env(256, 512)).cuda()
ctx = get_ctx(g["fx"], g["fy"])
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
outs, _, _ = RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 777)
torch.cuda.synchronize()
print(" ".join(hashlib.sha256(o.contiguous().cpu().numpy().tobytes()).hexdigest()[:12] for o in outs), "mean %.6f" % float(outs[0].mean()))
