import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR
v, t = M.scene.make_mesh(7, 64)
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda())
for _ in range(3): W.update_mesh(W.vrt, W.v_ind)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): W.update_mesh(W.vrt, W.v_ind)
b.record(); torch.cuda.synchronize()
print("LBVH build, T=%d: %.3f ms" % (len(t), a.elapsed_time(b) / 50))
