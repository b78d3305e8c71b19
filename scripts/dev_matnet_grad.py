import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D, GRADIENT_SCALING
mn, mx = M.scene.material_min_max(me_max=0.6)
mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=5)
with torch.no_grad(): mlp.encoder.params.mul_(2e3)
g = torch.Generator(device="cuda").manual_seed(3)
n = 60000
pts = torch.rand((n, 3), device="cuda", generator=g) * 1.6 - 0.8
w = torch.rand((n, 6), device="cuda", generator=g)
def loss(): return (mlp.sample(pts).double() * w).sum()
loss().backward()
P = mlp.encoder.params
for lv, (a, b) in enumerate([(0, 4096), (4096, 17920), (17920, 57224), (57224, 174880), (532792, 1057080)]):
    d = torch.zeros_like(P); d[2*a:2*b] = torch.randn(2*(b-a), device="cuda", generator=g)
    ana = float((P.grad.double() * d).sum()) / GRADIENT_SCALING
    base = P.detach().clone()
    res = []
    for eps in (0.0025, 0.01, 0.04):
        with torch.no_grad():
            P.copy_(base + eps * d); A = float(loss()); P.copy_(base - eps * d); B = float(loss()); P.copy_(base)
        res.append((A - B) / (2 * eps))
    print("level", lv, "ana", ana, "num", res)
