import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import dist as D
t = torch.arange(24, dtype=torch.float32, device="cuda").reshape(3, 8)
try:
    v = D.device_view(t.data_ptr(), (3, 8))
    print("view ok", v.shape, v.device, float(v.sum()), v.data_ptr() == t.data_ptr())
    v[1].zero_(); print(t[1].tolist())
except Exception as e:
    print("device_view failed:", type(e).__name__, e)
