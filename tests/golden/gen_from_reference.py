"""Generates tests/golden/ref_python.npz by IMPORTING the reference's own Python (this container only: /root/reference is absent on the GPU box).

What can be imported of the hot path: everything else is Slang->CUDA or needs slangpy/tinycudann/nvdiffrast.
  * nerf/render_dump.py:safe_l2_normalize            — used by run_restir_di_with_pt (renderer_restir.py:486)
  * nerf/ScreenSpaceReSTIR/GenerateLightTiles.py:make_sampleable — the torch half (cumsum / sum / normalisation / forced 1.0 entries) of the
    importance tables; its two kernel launches are served by a fake module `m` that runs this repo's oracle restatement of those kernels.
  * nerf/ScreenSpaceReSTIR/Denoising.py:EAWDenoise_use_phi / EAWDenoise_use_phi_no_di — the a-trous driver (iteration count, stepWidth /= 2
    with int() at the launch, ping-pong of the colour buffer); its kernel launches are served the same way, so the fixture pins the DRIVER
    (what mirres_render's finish and mirres-restir_nerf_mesh_amd/Denoising.py restate), not the kernel.
  * nerf/render_dump.py:dump_render / dump_render_run_mesh / GGX_specular / get_light_rgbs / batch_intersector and
    nerf/render_helper.py:generate_envir_map_dir — BASELINE configs[0] (64 x 64, 1 spp, direct lighting over the fixed lat-long light set, no
    ReSTIR): the reference's own code run end to end on CPU tensors, with a brute-force numpy Moeller-Trumbore loop (hits in front of the
    origin) as the `intersector` object the reference expects from outside (nerf/renderer.py:179 sets it to None) — nothing of this repo's
    engine or oracle is in the loop. render_helper.py cannot be imported (tinycudann / nvdiffrast at module top), so
    the one pure-torch function is compiled from the file's AST and executed as is. -> tests/golden/config1_dump_render.npz
  * nerf/renderer_restir.py:run_restir_di_with_pt (:473-550) — the frame's pre / post processing (occupancy threshold, averaging by the sample
    count, the five denoiser calls, kd (1 - metalness) D + S + I, background = 1, nan_to_num), executed from the file's AST (the module imports
    slangpy) with `restir_di_with_pt` replaced by a function that hands back prepared sums and the a-trous drivers being the reference's own
    Denoising.py over the oracle kernel. Pins what mirres_render's finish / mirres_render_finish restate.
  * nerf/renderer.py:scale_img_nhwc / scale_img_hwc (:61-76) — the --ssaa down-scale of the harness (plain bilinear, no antialiasing), by AST.
  * meshutils.py:auto_normals (+ length / dot / safe_normalize) — smooth vertex normals of the G-buffer front end, by AST
    (the module imports pymeshlab); `device='cuda'` of its fallback normal redirected to CPU.
  * nerf/utils.py:linear2srgb_torch (+ _clip_0to1_warn_torch) — the tone curve of the harness (SURVEY §8 a-H), taken by AST like above.
  * nerf/renderutils/ops.py:prepare_shading_normal(..., use_python=True) — the reference's own pure-torch validation path of its CUDA plugin
    (bsdf_prepare_shading_normal, :84-121; the module imports without building anything, _get_plugin() is never called): forward for the four flag
    combinations incl. zero-length inputs, and autograd gradients of all six inputs -> tests/golden/ref_shading_normal.npz, which pins the numpy
    restatement (oracle.prepare_shading_normal) and the HIP operator (csrc/normal.hip) — SURVEY §8 f-3.
The fixtures are data (inputs + outputs); no reference source text is stored.

    python tests/golden/gen_from_reference.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_function(path, name, namespace):
    """Execute ONE top-level function of a reference file that cannot be imported as a module (its unrelated imports are missing here)."""
    import ast
    tree = ast.parse(open(os.path.join(REF, path)).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    code = compile(ast.Module(body=[fn], type_ignores=[]), os.path.join(REF, path), "exec")
    exec(code, namespace)
    return namespace[name]


def shading_normal():
    """nerf/renderutils/ops.py through its use_python=True path (never _get_plugin()): forward (four flag combinations) + autograd gradients."""
    ops = load("nerf/renderutils/ops.py", "ref_renderutils_ops")
    rng = np.random.default_rng(2)
    shape = (1, 12, 17, 3)
    mk = lambda: rng.normal(size=shape).astype(np.float32)
    pos, sn, st, gn, pn = mk(), mk(), mk(), mk(), mk()
    pn[..., 2] = np.abs(pn[..., 2]); pn[0, 0, :4, 2] = -0.3                  # some negative z (clamped)
    sn[0, 1, 0] = 0; st[0, 1, 1] = 0                                          # zero-length inputs: F.normalize returns 0
    view = np.array([0.5, -2.0, 3.0], np.float32).reshape(1, 1, 1, 3)
    t = lambda a: torch.from_numpy(a.copy())
    out = {"pos": pos, "view": view, "perturbed": pn, "smooth_nrm": sn, "smooth_tng": st, "geom_nrm": gn}
    for two_sided in (True, False):
        for opengl in (True, False):
            o = ops.prepare_shading_normal(t(pos), t(view), t(pn), t(sn), t(st), t(gn), two_sided, opengl, use_python=True)
            out["fwd_%d%d" % (two_sided, opengl)] = o.numpy()
    default_p = np.array([0, 0, 1], np.float32).reshape(1, 1, 1, 3)            # what perturbed_nrm=None stands for (:148-149 builds it on 'cuda')
    out["fwd_default_perturbation"] = ops.prepare_shading_normal(t(pos), t(view), t(default_p), t(sn), t(st), t(gn), True, True, use_python=True).numpy()
    # gradients (rows 2.. : away from the zero-length rows, whose derivative torch defines through the eps clamp of F.normalize)
    w = rng.normal(size=(1, 10, 17, 3)).astype(np.float32)
    sl = (slice(None), slice(2, None))
    for two_sided, opengl in ((True, True), (False, False)):
        ins = [t(a[sl] if a.shape[1] > 1 else a).requires_grad_(True) for a in (pos, view, pn, sn, st, gn)]
        o = ops.prepare_shading_normal(*ins, two_sided, opengl, use_python=True)
        (o * t(w)).sum().backward()
        for name, a in zip(("pos", "view", "perturbed", "smooth_nrm", "smooth_tng", "geom_nrm"), ins):
            out["grad_%d%d_%s" % (two_sided, opengl, name)] = a.grad.numpy()
    out["grad_weight"] = w
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_shading_normal.npz"), **out)
    print("wrote ref_shading_normal.npz")


def config1(O, rd):
    """BASELINE configs[0]: the reference's dump_render on a 64 x 64 G-buffer of the synthetic mesh (subdivision 3: 1 280 + 128 triangles)."""
    import types
    import mirres_restir_nerf_mesh_amd as M
    rng = np.random.default_rng(3)
    v, t = M.scene.make_mesh(3, 8)
    info, aabb, _, _ = O.bvh_build(v, t)
    Himg = Wimg = 64
    eye, dirs = M.scene.camera_rays(Himg, Wimg)
    r = O.trace(info, aabb, v, t, O.make_rays(np.repeat(eye[None], Himg * Wimg, 0), dirs), True)
    mask = r["hit"] > 0
    pos = r["pos"][mask].astype(np.float32); nrm = r["normal"][mask].astype(np.float32); rays_d = dirs[mask].astype(np.float32)
    n = pos.shape[0]
    albedo = (0.2 + 0.6 * rng.random((n, 3))).astype(np.float32)
    rough = np.repeat((0.15 + 0.7 * rng.random((n, 1))).astype(np.float32), 3, 1)     # ks[:,1:2].repeat(1,3) (renderer.py:1134)
    fresnel = np.repeat((0.02 + 0.1 * rng.random((n, 1))).astype(np.float32), 3, 1)   # ks[:,2:3].repeat(1,3)
    env_h, env_w = 16, 32                                                            # --light_probe_res_hw
    env = M.scene.make_env(env_h, env_w).astype(np.float32)
    gen_dirs = load_function("nerf/render_helper.py", "generate_envir_map_dir", {"torch": torch, "np": np})
    lw, ld = gen_dirs(env_h, env_w)
    model = types.SimpleNamespace(light_area_weight=lw, fixed_viewdirs=ld)
    n_queries = [0]

    v0 = v[t[:, 0]].astype(np.float32); E1 = (v[t[:, 1]] - v[t[:, 0]]).astype(np.float32); E2 = (v[t[:, 2]] - v[t[:, 0]]).astype(np.float32)

    class BruteForceIntersector:   # what batch_intersector expects: intersects_closest(o, d, stream_compaction=True) -> (hit mask, ...)
        def intersects_closest(self, o, d, stream_compaction=True):
            o = o.numpy().astype(np.float32); d = d.numpy().astype(np.float32)
            n_queries[0] += o.shape[0]
            hit = np.zeros(o.shape[0], bool)
            with np.errstate(divide="ignore", invalid="ignore"):
                for c in range(0, o.shape[0], 4096):   # Moeller-Trumbore of every ray against every triangle, fp32
                    oc = o[c:c + 4096, None, :]; dc = d[c:c + 4096, None, :]
                    P = np.cross(dc, E2[None]); det = (E1[None] * P).sum(-1)
                    inv = np.float32(1) / det
                    Tv = oc - v0[None]
                    u = (Tv * P).sum(-1) * inv
                    Q = np.cross(Tv, E1[None])
                    vv = (dc * Q).sum(-1) * inv
                    tt = (E2[None] * Q).sum(-1) * inv
                    ok = (np.abs(det) >= 1e-15) & (u >= 0) & (u <= 1) & (vv >= 0) & (u + vv <= 1) & (tt > 0)
                    hit[c:c + 4096] = ok.any(1)
            return torch.from_numpy(hit), None, None, None, None, None

    T = lambda a: torch.from_numpy(a.copy())
    out = {}
    for method in ("stratified_sampling", "stratifed_sample_equal_areas"):
        rgb, diff, spec = rd.dump_render(BruteForceIntersector(), T(pos), T(nrm), T(albedo), T(rough), T(fresnel), T(rays_d), T(env), env_h, env_w, model,
                                         sample_method=method, color_chunk_size=1500, chunk_size=100000, device="cpu")
        k = "w" if method == "stratified_sampling" else "e"
        out["rgb_" + k] = rgb.numpy(); out["diff_" + k] = diff.numpy(); out["spec_" + k] = spec.numpy()
    spec_pairs = rd.GGX_specular(T(nrm[:64]), T(-rays_d[:64]), ld.reshape(1, -1, 3).repeat(64, 1, 1), T(rough[:64]), T(fresnel[:64])).numpy()
    light_rgbs = rd.get_light_rgbs(T(env), env_h, env_w, ld, device="cpu").numpy().reshape(-1, 3)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config1_dump_render.npz"), vert=v, tri=t, image_hw=np.array([Himg, Wimg], np.int32), mask=mask,
                        pos=pos, normal=nrm, rays_d=rays_d, albedo=albedo, rough=rough, fresnel=fresnel, env=env, env_hw=np.array([env_h, env_w], np.int32),
                        light_w=lw.numpy(), light_dirs=ld.numpy(), light_rgbs=light_rgbs, ggx_first64=spec_pairs, shadow_queries=np.int64(n_queries[0]), **out)
    print("wrote config1_dump_render.npz: %d surface points, %d lights, %d shadow queries, mean rgb %.4f" % (n, ld.shape[0], n_queries[0], float(out["rgb_w"].mean())))


def main():
    if "--only-shading-normal" in sys.argv:
        shading_normal()
        return
    from oracle import oracle as O
    rd = load("nerf/render_dump.py", "ref_render_dump")
    glt = load("nerf/ScreenSpaceReSTIR/GenerateLightTiles.py", "ref_glt")
    rng = np.random.default_rng(11)
    x = rng.normal(size=(64, 3)).astype(np.float32)
    x[0] = 0.0; x[1] = [1e-9, 0, 0]; x[2] = [3, -4, 0]
    norm_out = rd.safe_l2_normalize(torch.from_numpy(x), dim=-1).numpy()

    # make_sampleable: fake slang module whose kernels are the oracle's restatement; torch.zeros(device='cuda') redirected to CPU
    Hc, Wc = 12, 20
    env = (rng.random((Hc, Wc, 3)) * 2).astype(np.float32)
    env[2:5] = 0.0                                 # three adjacent black rows -> the middle one takes the uniform fallback (row_weight < 1e-4)
    tex = O.flip_env(env)

    class Launch:
        def __init__(self, fn): self.fn = fn
        def launchRaw(self, blockSize=None, gridSize=None): self.fn()

    class FakeM:
        def make_sampleable(self, env_tex, weight, width, height):
            def run():
                pdf = np.zeros(width * height, np.float32); cdf = np.zeros((width + 1) * height, np.float32)
                mp = np.zeros(height, np.float32); mc = np.zeros(height + 1, np.float32)
                # oracle kernel part only: un-normalised weights are what the Slang kernel writes (make_sampleable.slang:34-60)
                w = O.env_weights(env_tex.numpy(), width, height)
                weight.copy_(torch.from_numpy(w).reshape(-1, 1))
            return Launch(run)
        def Distribution2D(self, w, h, pdf_, cdf_):
            def run():
                p, c = O.distribution2d(pdf_.numpy().ravel().copy(), cdf_.numpy().ravel().copy(), w, h)
                pdf_.copy_(torch.from_numpy(p).reshape(pdf_.shape)); cdf_.copy_(torch.from_numpy(c).reshape(cdf_.shape))
            return Launch(run)

    real_zeros = torch.zeros
    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)
    torch.zeros = cpu_zeros
    try:
        pdf_, cdf_, mpdf_, mcdf_ = glt.make_sampleable(FakeM(), torch.from_numpy(tex), Wc, Hc)
    finally:
        torch.zeros = real_zeros
    # ---- EAW driver
    den = load("nerf/ScreenSpaceReSTIR/Denoising.py", "ref_denoising")
    fx, fy = 24, 20; Np = fx * fy
    occ = (rng.random((Np, 1)) > 0.2).astype(np.float32)
    col = rng.random((Np, 3)).astype(np.float32)
    # a gently curved sheet (the filter's normal / position weights must not vanish: phi = (2.0, 0.1, 0.001))
    yy, xx = np.meshgrid(np.arange(fy, dtype=np.float32), np.arange(fx, dtype=np.float32), indexing="ij")
    pos = np.stack([xx * 0.01, yy * 0.01, 0.02 * np.sin(xx * 0.3) * np.cos(yy * 0.2)], -1).reshape(Np, 3).astype(np.float32)
    nrm = np.stack([-0.3 * np.cos(xx * 0.3) * 0.2, 0.2 * np.sin(yy * 0.2) * 0.2, np.ones_like(xx)], -1).reshape(Np, 3).astype(np.float32)
    nrm += rng.normal(size=(Np, 3)).astype(np.float32) * 0.02; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    launches = []

    class FakeDen:
        def _run(self, PHI, framedim_x, framedim_y, stepWidth, occ_map, color, normal_map, pos_map, out_color):
            def run():
                launches.append(int(stepWidth))
                o = O.eaw(framedim_x, framedim_y, stepWidth, PHI[0], PHI[1], PHI[2], occ_map.numpy(), color.numpy(), normal_map.numpy(), pos_map.numpy())
                out_color.copy_(torch.from_numpy(np.asarray(o)).reshape(out_color.shape))
            return Launch(run)
        def process_EAWDenoise(self, **k): return self._run(**k)
        def process_EAWDenoise_no_di(self, **k): return self._run(**k)

    torch.zeros = cpu_zeros
    try:
        t = lambda a_: torch.from_numpy(a_.copy())
        eaw_di = den.EAWDenoise_use_phi(FakeDen(), 2.0, 0.1, 0.001, 2, 2, fx, fy, t(occ), t(col), t(nrm), t(pos)).numpy()
        eaw_nodi = den.EAWDenoise_use_phi_no_di(FakeDen(), 2.0, 0.1, 0.001, 4, 3, fx, fy, t(occ), t(col), t(nrm), t(pos)).numpy()
    finally:
        torch.zeros = real_zeros
    ns = {"torch": torch, "np": np}
    load_function("nerf/utils.py", "_clip_0to1_warn_torch", ns)
    l2s = load_function("nerf/utils.py", "linear2srgb_torch", ns)
    srgb_in = np.concatenate([np.array([-0.5, 0.0, 1e-7, 0.001, 0.0031308, 0.00313081, 0.5, 1.0, 1.7], np.float32), rng.random(55).astype(np.float32)])
    srgb_out = l2s(torch.from_numpy(srgb_in.copy())).numpy()
    # ---- run_restir_di_with_pt pre / post (the spp loop replaced by prepared sums)
    spp_f = 7
    sums = [(rng.random((Np, 3)) * spp_f * (0.5 + k)).astype(np.float32) for k in range(6)]   # total_color, diff, spec, color_1, diff_1, spec_1
    f_occ = rng.choice(np.array([0.0, 0.05, 0.3, 0.5, 0.7, 1.0], np.float32), size=(Np, 1)).astype(np.float32)
    f_kd = rng.random((Np, 3)).astype(np.float32); f_rm = rng.random((Np, 2)).astype(np.float32)
    f_rd = rng.normal(size=(Np, 3)).astype(np.float32)
    seen = {}

    def fake_loop(use_scale, sx, sy, sz, mlp_mat, worker, spp_, fxx, fyy, *rest):
        seen["occ_after_threshold"] = rest[-13].numpy().copy()      # occ_map as the loop receives it
        seen["ray_dir_norm"] = rest[-7].numpy().copy()              # ray_dir_map after safe_l2_normalize
        tt = [torch.from_numpy(a_.copy()) for a_ in sums]
        return tt[0], tt[3], tt[1], tt[2], tt[4], tt[5], torch.zeros(1), spp_
    ns2 = {"torch": torch, "np": np, "safe_l2_normalize": rd.safe_l2_normalize, "restir_di_with_pt": fake_loop,
           "EAWDenoise_use_phi": den.EAWDenoise_use_phi, "EAWDenoise_use_phi_no_di": den.EAWDenoise_use_phi_no_di}
    run_ref = load_function("nerf/renderer_restir.py", "run_restir_di_with_pt", ns2)
    torch.zeros = cpu_zeros
    try:
        t = lambda a_: torch.from_numpy(a_.copy())
        occ_t = t(f_occ)
        fin = run_ref(False, 1.0, 1.0, 1.0, None, None, None, *([None] * 7), FakeDen(), *([None] * 7), 128, 1024,
                      t(env), occ_t, t(nrm), torch.zeros(Np, 1), t(f_kd), t(f_rm), t(f_rd), t(pos), None, None, None, None,
                      fx, fy, spp_f, 2, 2, 2.0, 0.1, 0.001)
    finally:
        torch.zeros = real_zeros
    fin = [o.numpy() for o in fin]
    ns3 = {"torch": torch, "np": np}
    load_function("nerf/renderer.py", "scale_img_nhwc", ns3)
    scale_hwc = load_function("nerf/renderer.py", "scale_img_hwc", ns3)
    ssaa_in = rng.random((12, 16, 3)).astype(np.float32)
    ssaa_out2 = scale_hwc(torch.from_numpy(ssaa_in.copy()), (6, 8)).numpy()      # --ssaa 2
    ssaa_out4 = scale_hwc(torch.from_numpy(ssaa_in.copy()), (3, 4)).numpy()      # --ssaa 4
    import mirres_restir_nerf_mesh_amd as M
    an_v, an_t = M.scene.make_mesh(2, 4)
    an_v = np.concatenate([an_v, [[5.0, 5.0, 5.0]]]).astype(np.float32)        # an unreferenced vertex: the degenerate-normal fallback (0, 0, 1)
    ns4 = {"torch": torch, "np": np}
    for fn_ in ("length", "dot", "safe_normalize"):
        load_function("meshutils.py", fn_, ns4)
    auto_n = load_function("meshutils.py", "auto_normals", ns4)
    real_tensor = torch.tensor
    torch.tensor = lambda *a_, **k_: real_tensor(*a_, **{kk: vv for kk, vv in k_.items() if kk != "device"})
    try:
        an_out = auto_n(torch.from_numpy(an_v), torch.from_numpy(an_t.astype(np.int32)))[0].numpy()
    finally:
        torch.tensor = real_tensor
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_python.npz"), norm_in=x, norm_out=norm_out, env=env, srgb_in=srgb_in, srgb_out=srgb_out,
                        an_vert=an_v, an_tri=an_t.astype(np.int32), an_out=an_out,
                        ssaa_in=ssaa_in, ssaa_out2=ssaa_out2, ssaa_out4=ssaa_out4,
                        fin_spp=np.int32(spp_f), fin_sums=np.stack(sums), fin_occ=f_occ, fin_kd=f_kd, fin_rm=f_rm, fin_rd=f_rd, fin_out=np.stack(fin),
                        fin_occ_after=seen["occ_after_threshold"], fin_occ_inplace=occ_t.numpy(), fin_rd_norm=seen["ray_dir_norm"],
                        pdf=pdf_.numpy().ravel(), cdf=cdf_.numpy().ravel(), mpdf=mpdf_.numpy().ravel(), mcdf=mcdf_.numpy().ravel(),
                        eaw_dims=np.array([fx, fy], np.int32), eaw_occ=occ, eaw_col=col, eaw_nrm=nrm, eaw_pos=pos, eaw_di=eaw_di, eaw_nodi=eaw_nodi,
                        eaw_steps=np.array(launches, np.int32))
    print("wrote ref_python.npz")
    shading_normal()
    if "--skip-config1" not in sys.argv:
        config1(O, rd)   # ~4 min: 800 k rays against 1 408 triangles by brute force


if __name__ == "__main__":
    main()
