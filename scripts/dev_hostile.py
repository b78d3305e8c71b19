"""Dev (GPU box): the hostile-geometry traversal case of tests/test_gpu_bvh.py with per-category mismatch counts (which mode, which kind of ray)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle
from mirres_restir_nerf_mesh_amd.renderer_restir import restirbvhWorker
from mirres_restir_nerf_mesh_amd._lib import lib, check
rng = np.random.default_rng(21)
V, T = [], []
centres = (rng.random((24, 3)) * 200 - 100).astype(np.float64)
for c in centres:
    k = int(rng.integers(1, 7))
    for _ in range(k):
        e = rng.normal(size=(3, 3)) * rng.choice([1e-7, 1e-5, 1e-3])
        i = len(V); V.extend([c + e[0], c + e[1], c + e[2]]); T.append([i, i + 1, i + 2])
for _ in range(300):
    c = rng.random(3) * 200 - 100; e = rng.normal(size=(3, 3)) * 6.0
    i = len(V); V.extend([c, c + e[0], c + e[1]]); T.append([i, i + 1, i + 2])
v = np.array(V, np.float32); t = np.array(T, np.int32)
w = restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); w.update_mesh(w.vrt, w.v_ind)
info, aabb, _, _ = oracle.bvh_build(v, t)
n = 20000
o = (rng.random((n, 3)) * 240 - 120).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32)
tgt = v[t[rng.integers(0, len(t), size=n), 0]]
aim = rng.random(n) < 0.7
d[aim] = (tgt[aim] + rng.normal(size=(int(aim.sum()), 3)).astype(np.float32) * 1e-3) - o[aim]
o[0::17] = tgt[0::17] + (rng.normal(size=(len(o[0::17]), 3)) * 1e-6).astype(np.float32)
o[1::19] *= 8.0
d[2::23, 0] = np.float32(1e-40); d[3::23, 1] = np.float32(-1e-42); d[4::23, 2] = 0.0
d[5::29] = [1e-39, 1.0, 0.0]
cat = np.zeros(n, int); cat[0::17] = 1; cat[1::19] = 2; cat[2::23] = 3; cat[3::23] = 4; cat[4::23] = 5; cat[5::29] = 6
rays = oracle.make_rays(o, d)
ref = oracle.trace(info, aabb, v, t, rays, True, True)
dr = torch.from_numpy(rays).cuda()
for mode in (1, 2, 0):
    hit = torch.zeros(n, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, device="cuda"); pos = torch.zeros((n, 3), device="cuda")
    nrm = torch.zeros((n, 3), device="cuda"); pr = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, mode, hit.data_ptr(), tt.data_ptr() if mode else None, pos.data_ptr() if mode else None, nrm.data_ptr() if mode else None, pr.data_ptr() if mode else None, None, None), "m")
    torch.cuda.synchronize()
    bad = hit.cpu().numpy() != ref["hit"]
    badp = (pr.cpu().numpy() != ref["prim"]) if mode else bad
    print("mode", mode, "hit mismatches", int(bad.sum()), "prim mismatches", int(badp.sum()), "by category", np.bincount(cat[bad | badp], minlength=7).tolist())
    for i in np.nonzero(bad | badp)[0][:6]:
        print("   ray", i, "cat", cat[i], "o", o[i], "d", d[i], "ref hit/prim/t", ref["hit"][i], ref["prim"][i], ref["t"][i], "got", int(hit[i]), int(pr[i]), float(tt[i]))
# per-ray visit counters of the reference-order kernels (GPU, counted instantiation) against the oracle's for the first mismatching rays
cnt = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
hit = torch.zeros(n, dtype=torch.int32, device="cuda"); tt = torch.zeros(n, device="cuda"); pos = torch.zeros((n, 3), device="cuda"); nrm = torch.zeros((n, 3), device="cuda"); pr = torch.zeros(n, dtype=torch.int32, device="cuda")
check(lib().mirres_bvh_trace(w.h, dr.data_ptr(), n, 1, hit.data_ptr(), tt.data_ptr(), pos.data_ptr(), nrm.data_ptr(), pr.data_ptr(), cnt.data_ptr(), None), "c")
torch.cuda.synchronize()
bad = np.nonzero(hit.cpu().numpy() != ref["hit"])[0]
print("counted reference-order kernel: mismatches", len(bad), "counter keys", [k for k in ref.keys()])
c = cnt.cpu().numpy()
for i in bad[:5]:
    print("  ray", i, "gpu counters", c[i].tolist(), "oracle", [int(ref[k][i]) for k in ("popped", "entered", "leaves") if k in ref], "gpu hit", int(hit[i]), "t", float(tt[i]))
# the normalised direction and reciprocals as numpy float32 computes them
for i in bad[:3]:
    dd = d[i].astype(np.float32); l = np.float32(1) / np.sqrt(np.float32((dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2]))
    nd = dd * l
    with np.errstate(all="ignore"):
        print("  ray", i, "normalised d", nd, "1/d", np.float32(1) / np.where(nd == 0, np.float32(1e-6), nd))
oc = ref["counters"]
for i in bad[:5]:
    print("  ray", i, "oracle counters", oc[i].tolist())
