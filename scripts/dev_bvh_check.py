"""Dev check (GPU box): LBVH build + traversal parity against the oracle, plus a first traversal timing."""
import ctypes as C, sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
import importlib.util
spec = importlib.util.spec_from_file_location("scene", os.path.join(ROOT, "mirres-restir_nerf_mesh_amd", "scene.py")); S = importlib.util.module_from_spec(spec); spec.loader.exec_module(S)
L = C.CDLL(os.path.join(ROOT, "mirres-restir_nerf_mesh_amd", "libmirres.so"))
L.mirres_last_error.restype = C.c_char_p
vp = C.c_void_p
dev = "cuda"
def P(t): return vp(t.data_ptr()) if t is not None else None

def run(subdiv, gres, H, W, label):
    v, t = S.make_mesh(subdiv, gres)
    T = len(t)
    h = vp()
    assert L.mirres_bvh_create(C.byref(h), T) == 0
    dv = torch.from_numpy(v).to(dev); dt = torch.from_numpy(t).to(dev)
    info = torch.zeros((2*T-1, 3), dtype=torch.int32, device=dev); aabb = torch.zeros((2*T-1, 6), dtype=torch.float32, device=dev)
    srt = torch.zeros((T, 2), dtype=torch.int32, device=dev)
    rc = L.mirres_bvh_build(h, P(dv), len(v), P(dt), T, P(info), P(aabb), P(srt), None); assert rc == 0, L.mirres_last_error()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        L.mirres_bvh_build(h, P(dv), len(v), P(dt), T, P(info), P(aabb), P(srt), None)
    torch.cuda.synchronize(); bt = (time.time()-t0)/10
    t0 = time.time(); oi, oa, osrt, mh = O.bvh_build(v, t); ot = time.time()-t0
    print(f"[{label}] T={T} build gpu {bt*1e3:.3f} ms  oracle {ot*1e3:.1f} ms  height {mh}")
    print("  sorted equal:", np.array_equal(srt.cpu().numpy(), osrt), " info equal:", np.array_equal(info.cpu().numpy(), oi),
          " aabb equal:", np.array_equal(aabb.cpu().numpy(), oa))
    eye, rd = S.camera_rays(H, W)
    n = H*W
    rays = O.make_rays(np.repeat(eye[None], n, 0), rd)
    # secondary rays: from primary hits towards random directions
    r0 = O.trace(oi, oa, v, t, rays, True, True)
    rng = np.random.default_rng(1)
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    o2 = (r0["pos"] + 0.01 * d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    m = r0["hit"] > 0
    rays2 = O.make_rays(o2[m], d2[m])
    for name, rr in (("primary", rays), ("secondary", rays2)):
        k = len(rr)
        dr = torch.from_numpy(rr).to(dev)
        hit = torch.zeros(k, dtype=torch.int32, device=dev); tt = torch.zeros(k, device=dev); pos = torch.zeros((k, 3), device=dev)
        nrm = torch.zeros((k, 3), device=dev); prim = torch.zeros(k, dtype=torch.int32, device=dev); cnt = torch.zeros((k, 4), dtype=torch.int32, device=dev)
        rc = L.mirres_bvh_trace(h, P(dr), k, 1, P(hit), P(tt), P(pos), P(nrm), P(prim), P(cnt), None); assert rc == 0, L.mirres_last_error()
        torch.cuda.synchronize()
        ref = O.trace(oi, oa, v, t, rr, True, True)
        eq = lambda a, b: bool(np.array_equal(a, b))
        print(f"  closest/{name}: n={k} hit {eq(hit.cpu().numpy(), ref['hit'])} t {eq(tt.cpu().numpy().view(np.uint32), ref['t'].view(np.uint32))} "
              f"pos {eq(pos.cpu().numpy().view(np.uint32), ref['pos'].view(np.uint32))} normal {eq(nrm.cpu().numpy().view(np.uint32), ref['normal'].view(np.uint32))} "
              f"prim {eq(prim.cpu().numpy(), ref['prim'])} counters {eq(cnt.cpu().numpy().astype(np.uint32)[:, :3], ref['counters'][:, :3])} "
              f"hitfrac {ref['hit'].mean():.3f} popped/ray {ref['counters'][:,0].mean():.1f} leaves/ray {ref['counters'][:,2].mean():.2f}")
        hit2 = torch.zeros(k, dtype=torch.int32, device=dev)
        rc = L.mirres_bvh_trace(h, P(dr), k, 0, P(hit2), None, None, None, None, None, None); assert rc == 0
        torch.cuda.synchronize()
        print(f"  any/{name}: hit {eq(hit2.cpu().numpy(), ref['hit'])}")
        # timing
        for mode in (1, 0):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            L.mirres_bvh_trace(h, P(dr), k, mode, P(hit), P(tt), P(pos), P(nrm), P(prim), None, None)
            e0.record()
            for _ in range(5):
                L.mirres_bvh_trace(h, P(dr), k, mode, P(hit), P(tt), P(pos), P(nrm), P(prim), None, None)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)/5
            c = ref["counters"].astype(np.float64)
            bytes_ = (24 + 12 + 24*c[:, 0] + 24*c[:, 1] + 48*c[:, 2] + 28).sum()
            print(f"    mode {mode}: {ms:.3f} ms  {k/ms/1e6:.3f} Grays/s  ref-algorithmic {bytes_/ms/1e6:.1f} GB/s")
    L.mirres_bvh_destroy(h)

run(3, 16, 64, 64, "small")
run(5, 32, 256, 256, "medium")
if len(sys.argv) > 1:
    run(7, 64, 800, 800, "full")
