"""Generates tests/golden/ref_loop.npz: the REFERENCE's own frame loop — run_restir_di_with_pt -> restir_di_with_pt (nerf/renderer_restir.py:230-550),
the restirbvhWorker launch methods (:96-146) and the launch wrappers of nerf/ScreenSpaceReSTIR/{Resampling,GenerateLightTiles,Denoising}.py — executed
in this container (never on the GPU box) on CPU tensors, with the Slang module objects replaced by one fake module whose `process_*(...).launchRaw(...)`
run this repo's ORACLE kernels on exactly the tensors the reference hands over.

The BVH the loop traces comes from the reference's own `restirbvhWorker.update_bvh` (:25-89) executed the same way over the oracle's seven build
kernels (scene extent by torch min / max, `range(tree_heights.max())` refit passes, set_root), and is compared with the oracle's bvh_build.

The buffers come from the reference's own `load_m_for_restir` (:148-228) — `slangpy.loadModule` replaced by a recorder that returns the fake module
and keeps the `defines` each Slang file is compiled with (the ReSTIR constants), the neighbour-offset kernel served by the oracle.

`mlp_mat` is the reference's own `MLPTexture3D` / `_MLP` (nerf/render_helper.py:28-124, classes compiled from the AST: position normalisation and
clamp, torch.nn.Linear stack, sigmoid and range) with `tcnn.Encoding` replaced by the oracle's hash-grid encoder (tiny-cuda-nn is not in the image).

What this pins: the orchestration the oracle's orc_render (and, through it, mirres_render) restates — frame-index schedule (random_offset +
20 i + pass), the derived maps (normal_depth, brdf_map with its luminance weights / clamp / square), the two environment copies, reservoir
ping-pong, which buffers alias which across samples, the material look-ups between bounces, the nine accumulations, averaging, denoising,
compositing. What it cannot pin: the kernels themselves (Slang -> CUDA only).

renderer_restir.py cannot be imported (slangpy, pyexr, torchvision at module top), so the two functions and the three worker methods are compiled
from the file's AST and executed as they are. The fixture holds outputs only; the inputs are regenerated from seeds (tests/util.py:SmallFrame).

    python tests/golden/gen_reference_loop.py
"""
import ast
import ctypes as C
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"
SPP, SEED = 3, 4242
FRAME = dict(fx=48, fy=40, subdiv=3, ground=16, env_hw=(32, 64))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def ref_functions(path, names, namespace, cls=None):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    body = tree.body
    if cls:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    fns = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names), (names, [f.name for f in fns])
    exec(compile(ast.Module(body=fns, type_ignores=[]), os.path.join(REF, path), "exec"), namespace)
    return [namespace[n] for n in names]


def ref_classes(path, names, namespace):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert len(cls) == len(names)
    exec(compile(ast.Module(body=cls, type_ignores=[]), os.path.join(REF, path), "exec"), namespace)
    return [namespace[n] for n in names]


def reference_mlp(O, S, mat):
    """The reference's MLPTexture3D over the oracle's hash-grid encoder, seeded like matnet_for. Returns (module, the enc_cfg it asked tcnn for)."""
    import types
    seen = {}

    class Encoding:   # tcnn.Encoding(3, cfg): fp16 features [n,32]
        n_output_dims = 32
        def __init__(self, n_in, cfg): seen["n_in"], seen["cfg"] = n_in, dict(cfg)
        def register_full_backward_hook(self, fn): pass
        def __call__(self, x):
            bits = O.hashgrid_encode(mat, x.detach().numpy().astype(np.float32))
            return torch.from_numpy(bits.view(np.float16).reshape(-1, 32).copy())
    ns = {"torch": torch, "np": np, "tcnn": types.SimpleNamespace(Encoding=Encoding, free_temporary_memory=lambda: None)}
    _, MLPTexture3D = ref_classes("nerf/render_helper.py", ["_MLP", "MLPTexture3D"], ns)
    params, w0, w1, w2 = S.make_matnet_params(seed=0)
    mn, mx = S.material_min_max()
    tc, mc = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self; torch.nn.Module.cuda = lambda self, *a, **k: self   # the reference hard-codes .cuda(); no GPU here
    try:
        mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=[torch.from_numpy(mn), torch.from_numpy(mx)])
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = tc, mc
    with torch.no_grad():
        for i, w in zip((0, 2, 4), (w0, w1, w2)):
            mlp.net.net[i].weight.copy_(torch.from_numpy(w))
    return mlp, seen


def matnet_for(O, S):
    """Seeded material field (scene.make_matnet_params) as the oracle's struct + a stand-in for MLPTexture3D offering sample_no_di."""
    params, w0, w1, w2 = S.make_matnet_params(seed=0)
    mn, mx = S.material_min_max()
    keep = O.Keep()
    mat = O.matnet_struct(keep, params, w0, w1, w2, (-1, -1, -1), (1, 1, 1), mn, mx)

    class Mlp:
        calls = 0
        def sample_no_di(self, x):
            Mlp.calls += 1
            return torch.from_numpy(np.asarray(O.matnet(mat, x.numpy().astype(np.float32).reshape(-1, 3))).reshape(-1, 6))
    return mat, keep, Mlp()


def main():
    from oracle import oracle as O
    import mirres_restir_nerf_mesh_amd as M
    from util import SmallFrame
    S = M.scene
    F = SmallFrame(O, S, **FRAME)
    N, fx, fy = F.N, F.fx, F.fy
    mat, mkeep, mlp_standin = matnet_for(O, S)
    mlp, enc_seen = reference_mlp(O, S, mat)
    # the material field on its own: the reference's MLPTexture3D against the oracle's restatement
    rngp = np.random.default_rng(5)
    mat_pts = np.concatenate([(rngp.random((250, 3)) * 2.4 - 1.2), [[-1, -1, -1], [1, 1, 1], [0, 0, 0]]]).astype(np.float32)   # inside, outside (clamped), corners
    mat_ref = mlp.sample_no_di(torch.from_numpy(mat_pts)).numpy()
    mat_mine = np.asarray(O.matnet(mat, mat_pts)).reshape(-1, 6)
    print("MLPTexture3D (reference classes over the oracle encoder) vs orc_matnet: max |d| = %.3g; tcnn config asked: %s" % (float(np.abs(mat_ref - mat_mine).max()), enc_seen["cfg"]))
    L = O.lib()
    f32p, i32p, u64p = O.f32p, O.i32p, O.u64p
    log = []

    def A(t, dt=np.float32):   # the tensor's own memory (the kernels write in place)
        a = t.detach().numpy()
        assert a.dtype == dt and a.flags.c_contiguous, (a.dtype, a.shape)
        return a
    P = lambda t: A(t).ctypes.data_as(f32p)
    PI = lambda t: A(t, np.int32).ctypes.data_as(i32p)
    keepalive = []

    def dummy(n):
        z = np.zeros(n, np.float32); keepalive.append(z); return z.ctypes.data_as(f32p)

    def frame(kw):
        f = O.Frame(); f.fx, f.fy = int(kw["framedim_x"]), int(kw["framedim_y"]); n = f.fx * f.fy
        f.occ = P(kw["occ_map"]) if "occ_map" in kw else dummy(n)
        f.pos = P(kw["pos_map"]) if "pos_map" in kw else dummy(3 * n)
        f.normal_depth = P(kw["normal_depth"]) if "normal_depth" in kw else dummy(4 * n)
        f.brdf = P(kw["brdf_map"]) if "brdf_map" in kw else dummy(3 * n)
        f.ray_dir = P(kw["ray_dir"]) if "ray_dir" in kw else dummy(3 * n)
        if "g_lbvh_info" in kw:
            f.info = PI(kw["g_lbvh_info"]); f.aabb = P(kw["g_lbvh_aabb"]); f.vert = P(kw["vert"]); f.tri = PI(kw["v_indx"])
        if "env_tex" in kw:
            f.env_tex = P(kw["env_tex"]); f.env_w, f.env_h = int(kw["env_width"]), int(kw["env_height"])
        if "pdf_" in kw:
            f.pdf, f.cdf, f.mpdf, f.mcdf = P(kw["pdf_"]), P(kw["cdf_"]), P(kw["mpdf_"]), P(kw["mcdf_"])
        f.max_bounce = 2
        return f

    def res(r):
        s = O.Res(); s.light_data = P(r[0]); s.light_pdf = P(r[1]); s.M = PI(r[2]); s.weight = P(r[3]); return s

    class Launch:
        def __init__(self, fn, name): self.fn, self.name = fn, name
        def launchRaw(self, blockSize=None, gridSize=None):
            log.append(self.name); self.fn(gridSize)

    class FakeM:
        """One object stands for all eight Slang modules."""
        def make_sampleable(self, env_tex, weight, width, height):
            def run(_):
                weight.copy_(torch.from_numpy(O.env_weights(env_tex.numpy(), width, height)).reshape(-1, 1))
            return Launch(run, "make_sampleable")
        def Distribution2D(self, w, h, pdf_, cdf_):
            def run(_):
                p, c = O.distribution2d(pdf_.numpy().ravel().copy(), cdf_.numpy().ravel().copy(), w, h)
                pdf_.copy_(torch.from_numpy(p).reshape(pdf_.shape)); cdf_.copy_(torch.from_numpy(c).reshape(cdf_.shape))
            return Launch(run, "Distribution2D")
        def process_GenerateLightTiles(self, **k):
            def run(grid):
                f = O.Frame(); f.fx = f.fy = 1
                f.env_tex = P(k["env_tex"]); f.env_w, f.env_h = k["width"], k["height"]
                f.pdf, f.cdf, f.mpdf, f.mcdf = P(k["pdf_"]), P(k["cdf_"]), P(k["mpdf_"]), P(k["mcdf_"])
                L.orc_light_tiles(C.byref(f), C.c_uint32(k["frameIndex"]), int(grid[1]), int(grid[0]), P(k["light_data"]), PI(k["light_uv"]), P(k["light_inv_pdf"]))
            return Launch(run, "tiles@%d" % k["frameIndex"])
        def process_InitialResampling_(self, **k):
            def run(_):
                f = frame(k); r = res(k["reservoirs"])
                L.orc_initial(C.byref(f), C.byref(r), P(k["light_data"]), P(k["light_inv_pdf"]), C.c_uint32(k["frameIndex"]), None)
            return Launch(run, "initial@%d" % k["frameIndex"])
        def process_TemporalResampling(self, **k):
            def run(_):
                f = frame(k); r = res(k["reservoirs"]); p = res(k["prevReservoirs"])
                L.orc_temporal(C.byref(f), C.byref(r), C.byref(p), P(k["prev_occ_map"]), P(k["prev_normal_depth"]), P(k["prev_brdf_map"]), P(k["prev_ray_dir"]),
                               P(k["motionVectors"]), C.c_uint32(k["frameIndex"]))
            return Launch(run, "temporal@%d" % k["frameIndex"])
        def process_SpatialResampling_(self, **k):
            def run(_):
                f = frame(k); r = res(k["reservoirs"]); p = res(k["prevReservoirs"])
                L.orc_spatial(C.byref(f), C.byref(r), C.byref(p), P(k["neighborOffsets"]), C.c_uint32(k["frameIndex"]), None)
            return Launch(run, "spatial@%d" % k["frameIndex"])
        def process_EvaluateFinalSamples_get_vis(self, **k):
            def run(_):
                f = frame(k); r = res(k["reservoirs"])
                L.orc_final_vis(C.byref(f), C.byref(r), P(k["vis_map"]), None)
            return Launch(run, "final_vis")
        def process_EvaluateFinalSamples_di_(self, **k):
            def run(_):
                f = frame(k); r = res(k["reservoirs"]); fs = k["finalSample"]
                L.orc_eval_final(C.byref(f), C.byref(r), P(k["vis_map"]), P(fs[0]), P(fs[1]), P(fs[2]))
            return Launch(run, "eval_final")
        def process_FinalShading(self, **k):
            def run(_):
                kk = dict(k); kk["ray_dir"] = k["ray_dir"]
                f = frame(kk); fs = k["finalSample"]
                L.orc_final_shading(C.byref(f), P(k["normal"]), P(k["diffuse_map"]), P(k["linearRoughness_specular_map"]), P(fs[0]), P(fs[1]), P(fs[2]),
                                    P(k["color"]), P(k["diff_light"]), P(k["spec_light"]))
            return Launch(run, "final_shading")
        def _path(self, k):
            p = O.Path()
            for n, key in (("occ", "occ_map"), ("pos", "pos_map"), ("normal", "normal"), ("ray_dir", "ray_dir"), ("kd", "diffuse_map"), ("rs", "linearRoughness_specular_map"),
                           ("prd", "prd"), ("new_pos", "new_pos_map"), ("new_ray_d", "new_ray_d"), ("new_occ", "new_occ_map"), ("new_normal", "new_normal")):
                setattr(p, n, P(k[key]))
            return p
        def process_new_dir_for_pt(self, **k):
            def run(_):
                kk = {x: k[x] for x in ("framedim_x", "framedim_y", "g_lbvh_info", "g_lbvh_aabb", "vert", "v_indx")}
                f = frame(kk); p = self._path(k)
                L.orc_new_dir(C.byref(f), C.byref(p), C.c_uint32(k["frameIndex"]), C.c_uint32(k["bounce_count"]), None)
            return Launch(run, "new_dir@%d" % k["frameIndex"])
        def process_path_tracing_divided_no_grad(self, **k):
            def run(_):
                kk = {x: k[x] for x in ("framedim_x", "framedim_y", "g_lbvh_info", "g_lbvh_aabb", "vert", "v_indx", "env_tex", "env_width", "env_height", "pdf_", "cdf_", "mpdf_", "mcdf_")}
                f = frame(kk); p = self._path(k)
                L.orc_bounce(C.byref(f), C.byref(p), C.c_uint32(k["frameIndex"]), C.c_uint32(k["bounce_count"]), P(k["color"]), P(k["diff_color"]), P(k["spec_color"]), None)
            return Launch(run, "bounce%d@%d" % (k["bounce_count"], k["frameIndex"]))
        def process_EAWDenoise(self, **k): return self._eaw(**k)
        def process_EAWDenoise_no_di(self, **k): return self._eaw(**k)
        def _eaw(self, PHI, framedim_x, framedim_y, stepWidth, occ_map, color, normal_map, pos_map, out_color):
            def run(_):
                o = O.eaw(framedim_x, framedim_y, stepWidth, PHI[0], PHI[1], PHI[2], occ_map.numpy(), color.numpy(), normal_map.numpy(), pos_map.numpy())
                out_color.copy_(torch.from_numpy(np.asarray(o)).reshape(out_color.shape))
            return Launch(run, "eaw%d" % stepWidth)

    # ---- the reference's own Python
    rd = load("nerf/render_dump.py", "ref_render_dump")
    rs = load("nerf/ScreenSpaceReSTIR/Resampling.py", "ref_resampling")
    glt = load("nerf/ScreenSpaceReSTIR/GenerateLightTiles.py", "ref_glt")
    den = load("nerf/ScreenSpaceReSTIR/Denoising.py", "ref_denoising")
    ns = {"torch": torch, "np": np, "safe_l2_normalize": rd.safe_l2_normalize}
    for mod in (rs, glt, den):
        ns.update({k: v for k, v in vars(mod).items() if not k.startswith("_")})
    ref_functions("nerf/renderer_restir.py", ["restir_di_with_pt", "run_restir_di_with_pt"], ns)
    wns = dict(ns)
    methods = ref_functions("nerf/renderer_restir.py", ["InitialResampling_", "SpatialResampling_", "EvaluateFinalSamples_get_vis"], wns, cls="restirbvhWorker")

    build_log = []

    class BuildM:   # stands for the five bvhworkers modules
        def _l(self, name, fn):
            def run(_):
                build_log.append(name); fn()
            return Launch(lambda g: (build_log.append(name), fn()), name)
        def generateElements(self, vert, v_indx, ele_primitiveIdx, ele_aabb):
            return self._l("generateElements", lambda: L.orc_bvh_elements(P(vert), PI(v_indx), v_indx.shape[0], PI(ele_primitiveIdx), P(ele_aabb)))
        def pushConstantsMortonCodes(self, **k):
            import types
            return types.SimpleNamespace(**k)
        def morton_codes(self, pc, ele_aabb, morton_codes_ele):
            def fn():
                gmin = np.array([float(pc.g_min_x), float(pc.g_min_y), float(pc.g_min_z)], np.float32)
                gmax = np.array([float(pc.g_max_x), float(pc.g_max_y), float(pc.g_max_z)], np.float32)
                L.orc_bvh_morton(int(pc.g_num_elements), gmin.ctypes.data_as(f32p), gmax.ctypes.data_as(f32p), P(ele_aabb), PI(morton_codes_ele))
            return self._l("morton_codes", fn)
        def radix_sort(self, g_num_elements, g_elements_in, g_elements_out):
            return self._l("radix_sort", lambda: L.orc_bvh_radix_sort(g_num_elements, PI(g_elements_in), PI(g_elements_out)))
        def hierarchy(self, g_num_elements, ele_primitiveIdx, ele_aabb, g_sorted_morton_codes, g_lbvh_info, g_lbvh_aabb, g_lbvh_construction_infos):
            return self._l("hierarchy", lambda: L.orc_bvh_hierarchy(g_num_elements, PI(ele_primitiveIdx), P(ele_aabb), PI(g_sorted_morton_codes), PI(g_lbvh_info), P(g_lbvh_aabb),
                                                                      PI(g_lbvh_construction_infos)))
        def get_bvh_height(self, g_num_elements, g_lbvh_info, g_lbvh_aabb, g_lbvh_construction_infos, tree_heights):
            return self._l("get_bvh_height", lambda: L.orc_bvh_heights(g_num_elements, PI(g_lbvh_construction_infos), PI(tree_heights)))
        def get_bbox(self, g_num_elements, expected_height, g_lbvh_info, g_lbvh_aabb, g_lbvh_construction_infos):
            return self._l("get_bbox%d" % expected_height, lambda: L.orc_bvh_bbox_pass(g_num_elements, expected_height, PI(g_lbvh_info), P(g_lbvh_aabb), PI(g_lbvh_construction_infos)))
        def set_root(self, g_lbvh_info, g_lbvh_aabb):
            return self._l("set_root", lambda: L.orc_bvh_set_root(PI(g_lbvh_info), P(g_lbvh_aabb)))

    methods += ref_functions("nerf/renderer_restir.py", ["update_bvh", "update_mesh"], wns, cls="restirbvhWorker")

    class Worker:   # the attributes and methods of restirbvhWorker the frame uses (its __init__ only loads the Slang modules)
        pass
    for fn in methods:
        setattr(Worker, fn.__name__, fn)
    W = Worker()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a).copy())
    W.m_gen_ele = W.m_morton_codes = W.m_radixsort = W.m_hierarchy = W.m_bounding_box = BuildM()
    real_zeros0 = torch.zeros
    torch.zeros = lambda *a, **k: real_zeros0(*a, **{x: y for x, y in k.items() if x != "device"})
    try:
        W.update_mesh(T(F.vert), T(F.tri.astype(np.int32)))     # renderer.py:975
    finally:
        torch.zeros = real_zeros0
    bvh_same = bool(np.array_equal(W.LBVHNode_info.numpy(), F.info) and np.array_equal(W.LBVHNode_aabb.numpy(), F.aabb))
    print("update_bvh over the oracle's build kernels: %d launches (%s ... %s), identical to orc_bvh_build: %s" % (len(build_log), " ".join(build_log[:6]), build_log[-1], bvh_same))
    assert bvh_same
    del log[:]

    # the reference's own load_m_for_restir (renderer_restir.py:148-228)
    import time as _time, types as _types
    m = FakeM()
    defines = {}

    def createNeighborOffsetTexture(sampleCount, neighborOffsets):
        return Launch(lambda g_: L.orc_neighbor_offsets_raw(int(sampleCount), P(neighborOffsets)), "createNeighborOffsetTexture")
    m.createNeighborOffsetTexture = createNeighborOffsetTexture

    def loadModule(path, defines_=None, **k):
        defines.update(k.get("defines", defines_) or {})
        return m
    real_zeros, real_ones = torch.zeros, torch.ones
    strip = lambda f: (lambda *a, **k: f(*a, **{x: y for x, y in k.items() if x != "device"}))
    torch.zeros, torch.ones = strip(real_zeros), strip(real_ones)
    lns = {"torch": torch, "np": np, "time": _time, "slangpy": _types.SimpleNamespace(loadModule=lambda path, defines=None: loadModule(path, defines))}
    (load_ref,) = ref_functions("nerf/renderer_restir.py", ["load_m_for_restir"], lns)
    try:
        mods = load_ref(fx, fy)
    finally:
        torch.zeros, torch.ones = real_zeros, real_ones
    del log[:]
    assert all(x is m for x in mods[:8])
    light_data, light_uv, light_inv_pdf, reservoirs, prev_reservoirs, final_samples, noff, tile_count, tile_size = mods[8:17]
    layout = [(tuple(t_.shape), str(t_.dtype)) for t_ in (light_data, light_uv, light_inv_pdf, *reservoirs, *prev_reservoirs, *final_samples, noff)]
    print("load_m_for_restir: defines %s; tiles %d x %d; neighbour offsets max |d| vs the oracle %.3g" % (defines, tile_count, tile_size, float(np.abs(noff.numpy() - O.neighbor_offsets(8192).reshape(8192, 2)).max())))
    torch.zeros, torch.ones = strip(real_zeros), strip(real_ones)
    np.random.seed(SEED); random_offset = int(np.random.randint(2**20)); np.random.seed(SEED)   # what the loop will draw (:245)
    occ_in = T(F.occ).reshape(N, 1)
    try:
        outs = ns["run_restir_di_with_pt"](False, 1.0, 1.0, 1.0, mlp, None, W, m, m, m, m, m, m, m, m,
                                           light_data, light_uv, light_inv_pdf, reservoirs, prev_reservoirs, final_samples, noff, tile_count, tile_size,
                                           T(F.env), occ_in, T(F.normal), T(F.depth).reshape(N, 1), T(F.kd), T(F.rm), T(F.ray_dir_raw), T(F.pos),
                                           None, None, None, None, fx, fy, SPP, 2, 2, 2.0, 0.1, 0.001)
    finally:
        torch.zeros, torch.ones = real_zeros, real_ones
    outs = np.stack([o.numpy() for o in outs])
    # ---- the oracle's own restatement of the same loop
    mine = O.render(fx, fy, SPP, random_offset, (F.info, F.aabb), F.vert, F.tri, F.env, F.occ, F.normal, F.depth, F.kd, F.rm, F.ray_dir_raw, F.pos, mat=mat)
    names = ["final_color", "diffuse", "spec", "indirect", "indirect_diff", "indirect_spec"]
    worst = max(float(np.abs(outs[k] - mine[n]).max()) for k, n in enumerate(names))
    print("launches: %d (%s ...), random_offset %d" % (len(log), " ".join(log[:14]), random_offset))
    print("max |reference loop over oracle kernels - orc_render| = %.3g; mean final colour %.4f" % (worst, float(outs[0].mean())))
    np.savez_compressed(os.path.join(HERE, "ref_loop.npz"), outs=outs, spp=np.int32(SPP), seed=np.int32(SEED), random_offset=np.int64(random_offset),
                        mat_pts=mat_pts, mat_out=mat_ref, enc_per_level_scale=np.float64(enc_seen["cfg"]["per_level_scale"]),
                        enc_cfg=np.array([enc_seen["n_in"], enc_seen["cfg"]["n_levels"], enc_seen["cfg"]["n_features_per_level"], enc_seen["cfg"]["log2_hashmap_size"], enc_seen["cfg"]["base_resolution"]], np.int32),
                        defines_keys=np.array(sorted(defines)), defines_vals=np.array([int(defines[k_]) for k_ in sorted(defines)], np.int32),
                        buffer_layout=np.array(["%s %s" % l_ for l_ in layout]), neighbor_offsets=noff.numpy(), tile_count_size=np.array([tile_count, tile_size], np.int32),
                        launches=np.array(log), build_launches=np.array(build_log), bvh_info_crc=np.int64(int(np.bitwise_xor.reduce(W.LBVHNode_info.numpy().ravel().astype(np.int64) * np.arange(1, W.LBVHNode_info.numel() + 1)))),
                        frame=np.array([FRAME["fx"], FRAME["fy"], FRAME["subdiv"], FRAME["ground"], FRAME["env_hw"][0], FRAME["env_hw"][1]], np.int32),
                        occ_after=occ_in.numpy())
    print("wrote ref_loop.npz")


if __name__ == "__main__":
    main()
