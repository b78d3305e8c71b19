"""Dev (CPU, oracle only): what the reference's own LBVH traversal costs on the two synthetic meshes — tree depth, deepest stack, visits per ray — for primary
rays and for shadow-like rays (surface point -> cosine-ish hemisphere / sun direction).  python scripts/dev_mesh_stats.py [res]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mirres_restir_nerf_mesh_amd as M
from oracle import oracle as O
S = M.scene
res = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for name in ("icosphere", "clustered"):
    v, t = S.mesh_by_name(name)
    t0 = time.time(); info, aabb, _, _ = O.bvh_build(v, t); tb = time.time() - t0
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    eye, rd = S.camera_rays(res, res)
    rays = O.make_rays(np.repeat(eye[None], res * res, 0), rd)
    r = O.trace(info, aabb, v, t, rays, True, counters=True)
    fg = r["hit"] > 0
    rng = np.random.default_rng(0)
    n = int(fg.sum())
    rnd = rng.normal(size=(n, 3)); rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
    d = r["normal"][fg] + 0.98 * rnd; d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = r["pos"][fg] + 0.01 * d
    sh = O.make_rays(o.astype(np.float32), d.astype(np.float32))
    rs = O.trace(info, aabb, v, t, sh, False, counters=True)
    dp = O.trace_stack_depth(info, aabb, v, t, rays); ds = O.trace_stack_depth(info, aabb, v, t, sh)
    print("%-10s T=%d V=%d  tri area min/median/max %.2e/%.2e/%.2e (x%.0f)  LBVH depth %d (bound 30+ceil(log2 T) = %d)  oracle build %.2fs" %
          (name, len(t), len(v), area[area > 0].min(), np.median(area), area.max(), area.max() / area[area > 0].min(), O.tree_depth(info), 30 + int(np.ceil(np.log2(len(t)))), tb))
    print("   primary: foreground %.3f  visits/ray popped %.1f entered %.1f leaves %.2f  deepest stack %d" % ((fg.mean(),) + tuple(r["counters"][:, :3].mean(0)) + (int(dp.max()),)))
    print("   shadow : occluded   %.3f  visits/ray popped %.1f entered %.1f leaves %.2f  deepest stack %d  overflow %d" %
          ((rs["hit"].mean(),) + tuple(rs["counters"][:, :3].mean(0)) + (int(ds.max()), int(rs["counters"][:, 3].sum()))))
