"""GPU: the PRIVATE traversal layout (4-wide quantised nodes + 64-byte leaf records + the LDS-staged prefix) checked as a data structure, not through rays.

The shadow-ray and ordered closest-hit kernels are exact for EVERY ray iff (bvh_trace.hip "order-free any-hit traversal")
  (1) every one of the T leaf slots is reachable from node 0 exactly once (a missing leaf = a missed occluder, a doubled one is only waste),
  (2) every child box, decoded exactly as the kernel decodes it (fmaf(q, step, origin)), contains the exact LBVH boxes of all leaves below it,
  (3) a leaf record holds the reference's own leaf box and triangle of that slot (the test the reference applies: helperDi.slang:108-134, 172-195),
  (4) unused child entries point at the null leaf, whose inverted box no ray passes,
  (5) the LDS prefix is a copy of the first levels with the in-prefix references re-tagged.
Ray-based parity (test_gpu_bvh.py, test_gpu_clustered.py) samples rays; this covers the hierarchy builders (4-wide collapse of the reference LBVH, the extended-Morton
tree, its binned-SAH top built with device atomics) for all of them at once, on the icosphere, the lego-like mesh, the deep adversarial chain and tiny meshes."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

NODE = np.dtype([("org", "<f4", 3), ("step_x", "<f4"), ("qlo", "<u4", 3), ("qhi", "<u4", 3), ("step_y", "<f4"), ("step_z", "<f4"), ("ref", "<i4", 4)])
LEAF = np.dtype([("v0", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("lo", "<f4", 3), ("hi", "<f4", 3), ("prim", "<i4")])
TOPBIT = 0x20000000


def fetch_layout(W):
    from mirres_restir_nerf_mesh_amd._lib import lib, check
    L = lib(); L.mirres_debug_layout.argtypes = [C.c_void_p] * 4; L.mirres_debug_layout.restype = C.c_int
    T = int(W.v_ind.shape[0])
    nodes = np.zeros(T - 1, NODE); leaves = np.zeros(T + 1, LEAF); top = np.zeros(85, NODE)
    check(L.mirres_debug_layout(W.h, nodes.ctypes.data, leaves.ctypes.data, top.ctypes.data), "layout")
    return nodes, leaves, top


def decode(nodes, ids):
    """Child boxes [n, 4, 3] (lo, hi) of nodes `ids`, as bvh_trace.hip's node test sees them: fmaf(q, step, origin) — q * step is exact (8 bits times a
    power of two), so the float64 sum rounded to float32 IS the fused result."""
    nd = nodes[ids]
    step = np.stack([nd["step_x"], nd["step_y"], nd["step_z"]], 1).astype(np.float64)          # [n, 3]
    org = nd["org"].astype(np.float64)
    sh = (8 * np.arange(4, dtype=np.uint32))[None, :, None]
    qlo = ((nd["qlo"][:, None, :] >> sh) & 0xff).astype(np.float64)                              # [n, 4, 3]
    qhi = ((nd["qhi"][:, None, :] >> sh) & 0xff).astype(np.float64)
    lo = (qlo * step[:, None, :] + org[:, None, :]).astype(np.float32)
    hi = (qhi * step[:, None, :] + org[:, None, :]).astype(np.float32)
    return lo, hi, qlo, qhi


def check_layout(nodes, leaves, top, T, vert, tri, info, aabb):
    """Returns statistics; raises AssertionError on any structural fault."""
    # ---- (3) leaf records = the reference's leaf nodes (LBVHNode_info[T-1+slot][2] = element id, LBVHNode_aabb[T-1+slot] = its box)
    prim = info[T - 1:, 2]
    assert np.array_equal(leaves["prim"][:T], prim)
    assert np.array_equal(np.sort(prim), np.arange(T)), "the sorted element ids are a permutation"
    assert np.array_equal(leaves["lo"][:T], aabb[T - 1:, 0:3]) and np.array_equal(leaves["hi"][:T], aabb[T - 1:, 3:6])
    a, b, c = vert[tri[prim, 0]], vert[tri[prim, 1]], vert[tri[prim, 2]]
    assert np.array_equal(leaves["v0"][:T], a) and np.array_equal(leaves["e1"][:T], b - a) and np.array_equal(leaves["e2"][:T], c - a)
    # the null leaf: inverted on every axis
    assert leaves["prim"][T] == -1 and np.all(leaves["lo"][T] > leaves["hi"][T])
    # ---- (1) breadth-first from node 0
    levels = [np.array([0], np.int64)]
    seen_node = np.zeros(max(1, T - 1), np.int32); seen_node[0] = 1
    seen_leaf = np.zeros(T + 1, np.int64)
    n_unused = 0
    while True:
        cur = levels[-1]
        ref = nodes["ref"][cur].astype(np.int64)                                     # [n, 4]
        lo, hi, qlo, qhi = decode(nodes, cur)
        isleaf = ref < 0
        slots = ~ref[isleaf]
        assert np.all(slots <= T)
        np.add.at(seen_leaf, slots, 1)
        # ---- (4) unused entries: null leaf <=> (lo = 255 > hi = 0 on every axis)
        null = isleaf & (~ref == T)
        marked = np.all(qlo == 255, axis=2) & np.all(qhi == 0, axis=2)
        assert np.array_equal(null, marked), "unused entries and the null leaf go together"
        assert np.all(null[:, 0] == False) and np.all(null[:, 1] == False), "a node has at least two children"
        n_unused += int(null.sum())
        kids = ref[~isleaf]
        assert np.all(kids < T - 1)
        np.add.at(seen_node, kids, 1)
        assert np.all(seen_node[kids] == 1), "a node is referenced once"
        if kids.size == 0:
            break
        levels.append(kids)
        assert len(levels) < 400
    assert np.all(seen_leaf[:T] == 1), "every leaf slot is reachable exactly once (missing %d, doubled %d)" % (int((seen_leaf[:T] == 0).sum()), int((seen_leaf[:T] > 1).sum()))
    # ---- (2) containment, bottom-up: exact union of the leaf boxes below every reachable node
    ex_lo = np.full((max(1, T - 1), 3), np.inf, np.float32); ex_hi = np.full((max(1, T - 1), 3), -np.inf, np.float32)
    worst = 0.0
    for cur in reversed(levels):
        ref = nodes["ref"][cur].astype(np.int64)
        lo, hi, _, _ = decode(nodes, cur)
        isleaf = ref < 0
        slot = np.where(isleaf, ~ref, 0); kid = np.where(isleaf, 0, ref)
        c_lo = np.where(isleaf[..., None], leaves["lo"][slot], ex_lo[kid])          # [n, 4, 3] exact boxes of the four children
        c_hi = np.where(isleaf[..., None], leaves["hi"][slot], ex_hi[kid])
        real = ~(isleaf & (slot == T))
        ok = (lo <= c_lo) & (hi >= c_hi)
        assert np.all(ok[real]), "a decoded child box does not contain its subtree (%d faults)" % int((~ok[real]).sum())
        ex_lo[cur] = np.min(np.where(real[..., None], c_lo, np.inf), axis=1)
        ex_hi[cur] = np.max(np.where(real[..., None], c_hi, -np.inf), axis=1)
        # how loose the 8-bit boxes are (reported, not asserted): relative to the node's extent
        ext = np.maximum((ex_hi[cur] - ex_lo[cur]).max(axis=1).astype(np.float64), 1e-30)[:, None, None]
        loose = np.where(real[..., None], np.maximum(c_lo.astype(np.float64) - lo, hi.astype(np.float64) - c_hi) / ext, 0.0)
        worst = max(worst, float(loose.max()))
    assert np.array_equal(ex_lo[0], aabb[0, 0:3]) and np.array_equal(ex_hi[0], aabb[0, 3:6]), "the union of everything below node 0 is the reference's root box"
    # ---- (5) the LDS prefix (built when T - 1 >= 1364): heap order, entry e's k-th child at 4e + 1 + k
    if T - 1 >= 341 * 4:
        ids = np.full(85, -1, np.int64); ids[0] = 0
        for e in range(85):
            if ids[e] < 0:
                assert np.all(top["ref"][e] == ~T) and np.all(top["qlo"][e] == 0xffffffff) and np.all(top["qhi"][e] == 0)
                continue
            src = nodes[ids[e]]
            for f in ("org", "step_x", "step_y", "step_z", "qlo", "qhi"):
                assert np.array_equal(top[f][e], src[f]), (e, f)
            for k in range(4):
                r, cs = int(src["ref"][k]), 4 * e + 1 + k
                if r >= 0 and cs < 85:
                    assert int(top["ref"][e][k]) == (TOPBIT | cs); ids[cs] = r
                else:
                    assert int(top["ref"][e][k]) == r
    return {"levels": len(levels), "reachable_nodes": int(sum(len(l) for l in levels)), "unused_entries": n_unused, "loosest_box_over_node_extent": worst}


def _worker(v, t, upgrade=True):
    import torch
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR
    W = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
    if upgrade:
        W.upgrade()      # round 6: update_mesh stops after the extended-Morton tree; the binned-SAH top is added when a frame is long enough — or here (a no-op when MIRRES_PRIVATE_TREE fixes the level)
    return W


def _meshes(scene_mod):
    from test_oracle_invariants import adversarial_chain_mesh
    yield "icosphere subdiv 7", scene_mod.mesh_by_name("icosphere")
    yield "lego-like, full size", scene_mod.mesh_by_name("clustered")
    yield "lego-like, small", scene_mod.make_mesh_clustered(target_tris=20000, seed=3)
    yield "adversarial chain (deep tree, duplicates)", adversarial_chain_mesh(dups=40)[:2]
    rng = np.random.default_rng(5)
    for T in (2, 3, 7, 8, 9, 64, 1365, 1366):        # around T >= 8 (private tree on) and T - 1 >= 1364 (LDS prefix on)
        v = rng.uniform(-1, 1, (3 * T, 3)).astype(np.float32)
        v[1::3] = v[0::3] + rng.normal(0, 0.05, (T, 3)).astype(np.float32); v[2::3] = v[0::3] + rng.normal(0, 0.05, (T, 3)).astype(np.float32)
        yield "soup of %d" % T, (v, np.arange(3 * T, dtype=np.int32).reshape(T, 3))


def run_all(scene_mod, oracle, upgrade=True):
    out = []
    for name, (v, t) in _meshes(scene_mod):
        W = _worker(v, t, upgrade)
        T = int(t.shape[0])
        info, aabb = W.LBVHNode_info.cpu().numpy(), W.LBVHNode_aabb.cpu().numpy()
        o_info, o_aabb, _, _ = oracle.bvh_build(v, t)
        assert np.array_equal(info, o_info) and np.array_equal(aabb, o_aabb)
        nodes, leaves, top = fetch_layout(W)
        st = check_layout(nodes, leaves, top, T, v, t, info, aabb)
        out.append("%-44s T %7d: %s" % (name, T, st))
    return out


def test_private_layout_is_a_valid_hierarchy(scene_mod, oracle):
    """Default configuration of this process (MIRRES_PRIVATE_TREE unset): what update_mesh builds (the extended-Morton tree) AND what a long frame upgrades it to
    (the same tree with the binned-SAH top, mirres_bvh_upgrade)."""
    from mirres_restir_nerf_mesh_amd._lib import lib
    lines = run_all(scene_mod, oracle, upgrade=False) + run_all(scene_mod, oracle, upgrade=True)
    if os.environ.get("MIRRES_PRIVATE_TREE") in (None, "", "auto"):
        W0 = _worker(*scene_mod.mesh_by_name("icosphere"), upgrade=False)
        assert lib().mirres_bvh_private_level(W0.h) == 1
        W0.ensure_hierarchy_for(1.0e7); assert lib().mirres_bvh_private_level(W0.h) == 1          # a short frame keeps the cheap tree
        W0.ensure_hierarchy_for(1.0e9); assert lib().mirres_bvh_private_level(W0.h) == 2          # a long one completes it
    rep = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(rep):
        with open(os.path.join(rep, "layout_check.txt"), "a") as f:
            f.write("MIRRES_PRIVATE_TREE=%s\n" % os.environ.get("MIRRES_PRIVATE_TREE", "(default 2)") + "\n".join(lines) + "\n")


@pytest.mark.parametrize("mode", ["0", "1"])
def test_the_other_hierarchies_are_valid_too(mode):
    """MIRRES_PRIVATE_TREE is read once per process: 0 = 4-wide collapse of the reference LBVH, 1 = extended-Morton tree without the SAH top."""
    env = dict(os.environ, MIRRES_PRIVATE_TREE=mode)
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "layout ok" in r.stdout


if __name__ == "__main__":
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import mirres_restir_nerf_mesh_amd as M
    from oracle import oracle as O
    O.lib()
    scene_mod = M.scene
    for line in run_all(scene_mod, O):
        print(line)
    print("layout ok")
