"""Dev (GPU box): GEMM phase of the material MLP in isolation (mirres_matnet_mlp on precomputed encodings) + the fused scatter path."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
mn, mx = M.scene.material_min_max()
mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=1)
with torch.no_grad(): mlp.encoder.params.mul_(1e3)
n = 2560000
g = torch.Generator(device="cuda").manual_seed(0)
pts = torch.rand((n, 3), device="cuda", generator=g) * 1.2 - 0.6
enc = mlp.encode(pts)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t_mlp = timeit(lambda: mlp.mlp_on_encoding(enc))
t_full = timeit(lambda: mlp.sample_no_di(pts))
alg = 4480.0 * n; issued = 32 * 2 * 32 * 32 * 16 / 64.0 * n
print(f"mlp_mfma (GEMM phase): {t_mlp:.3f} ms  algorithmic {alg/t_mlp/1e9:.1f} TFLOP/s  issued-MFMA {issued/t_mlp/1e9:.1f} TFLOP/s ({100*issued/t_mlp/1e9/2500:.1f}% of 2.5 PF f16 dense)  bytes {n*88/t_mlp/1e6:.0f} GB/s")
print(f"valu encode+mlp kernel: {t_full:.3f} ms")
