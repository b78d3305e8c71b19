import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
from mirres_restir_nerf_mesh_amd._ops import get_ctx
S = M.scene
res, spp = int(sys.argv[1]), int(sys.argv[2])
v, t = S.make_mesh(5, 16)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
g = harness.build_gbuffer(W, res, res, 1)
fx, fy = g["fx"], g["fy"]; N = fx * fy
env = torch.full((256, 512, 3), 0.5, device="cuda")
ctx = get_ctx(fx, fy)
tape = torch.full((spp * N, 8), -7.0, device="cuda")
sums, a, keep = RR.render_fused(ctx, W, None, False, (1, 1, 1), env, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp, 0, 1, 1.0, 1.0, 1.0, 4242,
                                spp_range=(0, spp), tape=tape)
torch.cuda.synchronize()
tp = tape.view(spp, N, 8)
for s in range(spp):
    x = tp[s]
    print("sample %2d: unwritten %7d  valid %7d  visible %7d  weight sum %.4e" % (s, int((x[:, 0] == -7).sum()), int((x[:, 0] > 0.1).sum()), int(((x[:, 0] > 0.1) & (x[:, 6] > 0)).sum()), float(x[:, 5].sum())))
# linearity of mirres_render_bwd over the samples of the tape
import ctypes as C
from mirres_restir_nerf_mesh_amd._lib import lib, check, stream_ptr
gc = torch.rand((N, 3), device="cuda"); gd = torch.rand((N, 3), device="cuda"); gs = torch.rand((N, 3), device="cuda")
def bwd(tp_, S):
    a.tape = tp_.data_ptr()
    gn = torch.empty((N, 3), device="cuda"); gk = torch.empty((N, 3), device="cuda"); gr = torch.empty((N, 2), device="cuda"); ge = torch.zeros((256, 512, 3), device="cuda")
    check(lib().mirres_render_bwd(ctx.h, C.byref(a), S, gc.data_ptr(), gd.data_ptr(), gs.data_ptr(), gn.data_ptr(), gk.data_ptr(), gr.data_ptr(), ge.data_ptr(), stream_ptr()), "bwd")
    torch.cuda.synchronize()
    return gn, gk, gr, ge
full = bwd(tape, spp)
parts = [bwd(tape[s * N:(s + 1) * N].contiguous(), 1) for s in range(spp)]
for i, nm in enumerate(("normal", "kd", "rm", "env")):
    ssum = sum(p[i].double() for p in parts)
    print("%6s: |full| %.5e |sum of per-sample| %.5e rel diff %.3e" % (nm, float(full[i].double().norm()), float(ssum.norm()), float((full[i].double() - ssum).norm() / (ssum.norm() + 1e-30))))
    print("        per-sample norms", ["%.3e" % float(p[i].norm()) for p in parts])
