"""Dev (GPU box): directional finite difference of the stage-1 loss w.r.t. the vertex offsets against autograd through harness.render_stage1_outputs
(interpolate -> material field position gradient, normals -> shading backward, dr.antialias), sampling seeds pinned."""
import sys, os, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, losses, raster, checkpoint as CK
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
from train_stage1 import orbit_pose
torch.manual_seed(0)
v, t = M.scene.make_mesh(4, 8)
verts = torch.from_numpy(v).cuda(); tris = torch.from_numpy(t).cuda()
aabb, mn, mx = CK.material_field_args(CK.material_config(bound=1.0))
mlp = MLPTexture3D(aabb, channels=6, min_max=(mn.cuda(), mx.cuda()), seed=3)
with torch.no_grad(): mlp.encoder.params.mul_(2e3)
env = torch.from_numpy(M.scene.make_env(32, 64)).cuda()
H = Wd = 64
Wk = RR.restirbvhWorker(verts, tris); Wk.update_mesh(Wk.vrt, Wk.v_ind)
mods = RR.load_m_for_restir(Wd, H)
topo = raster.antialias_topology(tris)
pose = torch.from_numpy(orbit_pose(30, 30)); focal = 0.5 * Wd / np.tan(0.5 * 0.6911); intr = (focal, focal, Wd * 0.5, H * 0.5)
gt = torch.full((H * Wd, 3), 0.7, device="cuda"); gt_lin = gt ** 2.2
opt = types.SimpleNamespace(use_brdf=True, lambda_lap=0.0, lambda_offsets=0.0)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
def loss_of(voff, spp=16):
    RR.set_random_offset(1234); torch.manual_seed(7)
    out = harness.render_stage1_outputs(Wk, verts, voff, tris, mlp, env, mods, H, Wd, spp, pose=pose, intrinsics=intr, topology=topo)
    if which == "alpha": return out["occ"].sum() * 1e-3
    if which == "image": return ((out["image_brdf"] - gt) ** 2).mean()
    return losses.stage1_loss(out, gt, gt_lin, opt, vertices=verts, voffsets=voff, triangles=tris)
voff = torch.zeros_like(verts).requires_grad_(True)
L0 = loss_of(voff); (g,) = torch.autograd.grad(L0, voff)
print("loss", float(L0), "grad norm", float(g.norm()), "nonzero", int((g.abs().sum(1) > 0).sum()), "of", len(g))
gen = torch.Generator(device="cuda").manual_seed(1)
for trial in range(4):
    # smooth direction: a low-frequency displacement field (radial scaling + shear), so that no discrete visibility event dominates
    c = torch.randn(3, 3, device="cuda", generator=gen) * 0.5
    d = verts @ c.t() + torch.randn(3, device="cuda", generator=gen) * 0.2
    for eps in (1e-3, 3e-4):
        with torch.no_grad():
            lp = float(loss_of((eps * d).contiguous())); lm = float(loss_of((-eps * d).contiguous()))
        print("trial %d eps %.0e: FD %.6e   autograd %.6e" % (trial, eps, (lp - lm) / (2 * eps), float((g * d).sum())))
