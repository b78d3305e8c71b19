"""Dev (GPU box): how often the shadow-ray kernel pays for its leaf branch (mirres_ctx_stats [13..15], counting kernel) on the frame's own rays, both meshes.
    python scripts/dev_leaf_branch.py [spp=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._ops import get_ctx
import bench as B
S = M.scene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
for mesh in ("icosphere", "clustered"):
    v, t = S.mesh_by_name(mesh)
    W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
    mlp = B.make_field(S, torch, dev)
    g = harness.build_gbuffer(W, 800, 800, 2, mlp_mat=mlp)
    env = torch.from_numpy(S.make_env(256, 512)).cuda()
    ctx = get_ctx(g["fx"], g["fy"])
    ctx.set_instrument(1); ctx.stats(reset=True)
    RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp, 2, 2, 2.0, 0.1, 0.001, 12345)
    st = ctx.stats(reset=True); ctx.set_instrument(0)
    it, lit, lv, rec, rays = st["any_wave_iters"], st["any_wave_leaf_iters"], st["any_leaf_visits"], st["entered"], st["rays_any"]
    print("%-10s rays %d: wave iterations %d (%.1f lane-records each = busy lanes), leaf branch run in %.1f %% of them for %.2f lanes on average; "
          "leaf records %.3f per ray = %.1f %% of the %.2f records per ray; exact-box passes (triangle tests) %.3f per ray" %
          (mesh, rays, it, rec / max(1, it), 100.0 * lit / max(1, it), lv / max(1, lit), lv / max(1, rays), 100.0 * lv / max(1, rec), rec / max(1, rays), st["leaves"] / max(1, rays)))
