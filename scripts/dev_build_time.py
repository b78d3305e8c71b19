import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR
v, t = M.scene.mesh_by_name(os.environ.get('MIRRES_MESH', 'icosphere'))
W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda())
for _ in range(3): W.update_mesh(W.vrt, W.v_ind)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): W.update_mesh(W.vrt, W.v_ind)
b.record(); torch.cuda.synchronize()
print("LBVH build, T=%d: %.3f ms" % (len(t), a.elapsed_time(b) / 50))
import ctypes as C
from mirres_restir_nerf_mesh_amd._lib import lib
L = lib()
try:
    L.mirres_debug_sah_state.argtypes = [C.c_void_p, C.c_void_p]; L.mirres_debug_sah_state.restype = C.c_int
    st = (C.c_uint32 * 8)(); L.mirres_debug_sah_state(W.h, st)
    print("SAH top: clusters %d, nodes above the cut %d, rebuilt nodes %d (internal %d), resolved %d, fail %d, levels %d" % tuple(st[:7]))
except AttributeError:
    pass
