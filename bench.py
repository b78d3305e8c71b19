#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json on MI355X: Msamples/s (pixels x spp) of the ReSTIR + path-tracing forward render.

    python bench.py [--gpus N --steps K --warmup W] [--res 800 --ssaa 2 --spp 512 --bounces 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one frame of the hot path: LBVH rebuild (restirbvhWorker.update_mesh, as render_stage1 does every frame) + run_restir_di_with_pt
(spp loop: light tiles, initial / temporal / spatial reservoir passes, final shading, 2 indirect bounces with the material network, EAW
denoise, composite) over a synthetic scene of BASELINE config 2's shape, with every input already resident in HBM. The G-buffer (primary
visibility, nvdiffrast's job in the reference) is built once outside the timed region.

N > 1: the one frame is shared by the ranks (mirres-restir_nerf_mesh_amd/dist.py) — total work is fixed, i.e. STRONG scaling, as BASELINE's "at
1/2/4/8 GPU" — and BOTH sharding schemes are timed in the same run: `value` = spp slices + one all-reduce (RCCL) of the accumulators, and the
`strips` sub-record = the north-star's pixel split (row strips, per-sample reservoir halo exchange, all-gather of radiance rows; bit-identical to one GPU).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (traversal kernel, measured live with HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--res", type=int, default=800, help="output resolution (BASELINE config 2: 800)")
    p.add_argument("--ssaa", type=int, default=2, help="reference default --ssaa 2 -> internal 1600x1600 (main.py:140)")
    p.add_argument("--spp", type=int, default=512, help="BASELINE.json metric: 800x800, 512 spp, 3-bounce (configs[1] quotes the same frame at 128 spp)")
    p.add_argument("--bounces", type=int, default=2, help="indirect bounces (MAX_Bounce, FinalShading.slang:7) -> 3 path vertices")
    p.add_argument("--subdiv", type=int, default=7, help="icosphere subdivisions (7 -> 327 680 + 8 192 ground triangles)")
    p.add_argument("--shard", choices=("both", "strips", "spp"), default="both",
                   help="N > 1: `spp` = sample slices + one all-reduce (the timed `value`), `strips` = exact row strips + per-sample halo exchange + all-gather (the north-star's "
                        "tile split, reported as the `strips` sub-record), `both` = time the two schemes one after the other")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--const-material", action="store_true", help="constant material instead of the hash-grid + MLP field")
    return p.parse_args()


def cpu_baseline(S, args):
    """The CPU oracle (restatement of the reference kernels — NOT reference code, which is CUDA-only) on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as O
    v, t = S.make_mesh(args.subdiv, 64 if args.subdiv >= 6 else 16)
    info, aabb, _, _ = O.bvh_build(v, t)
    fx = fy = 320
    eye, rd = S.camera_rays(fy, fx)
    r = O.trace(info, aabb, v, t, O.make_rays(np.repeat(eye[None], fx * fy, 0), rd), True)
    occ = r["hit"].astype(np.float32)
    nrm = np.where(occ[:, None] > 0, r["normal"], 0).astype(np.float32)
    depth = np.linalg.norm(r["pos"] - eye, axis=1).astype(np.float32)
    N = fx * fy
    kd = np.full((N, 3), 0.6, np.float32); rm = np.zeros((N, 2), np.float32); rm[:, 0] = 0.5
    mat = None
    keep = O.Keep()
    if not args.const_material:
        params, w0, w1, w2 = S.make_matnet_params(seed=0)
        mn, mx = S.material_min_max()
        mat = O.matnet_struct(keep, params, w0, w1, w2, (-1, -1, -1), (1, 1, 1), mn, mx)
        km = O.matnet(mat, r["pos"])
        kd = km[:, 0:3].copy(); rm = km[:, 4:6].copy()
    env = S.make_env(256, 512)
    last = {}
    def run(n_spp):
        t0 = time.time()
        O.bvh_build(v, t)
        last["out"] = O.render(fx, fy, n_spp, 12345, (info, aabb), v, t, env, occ, nrm, depth, kd, rm, rd, r["pos"], mat=mat, max_bounce=args.bounces)
        return time.time() - t0
    cal = run(1)                                           # calibration pass, then size the sample for ~15 s of CPU work
    spp = int(max(2, min(256, round(15.0 / max(cal, 1e-3)))))
    dt = run(spp)
    frame = dict(fx=fx, fy=fy, spp=spp, occ=occ, normal=nrm, depth=depth, kd=kd, rm=rm, ray_dir=rd, pos=r["pos"], env=env, final_color=last["out"]["final_color"])
    return {"value": round(N * spp / dt / 1e6, 6), "unit": "Msamples/s", "cores": O.num_threads(), "kind": "port",
            "sample": "%dx%d px, %d spp, same mesh (T=%d) / env / material, LBVH build + full frame (CPU restatement of the reference kernels, not reference code); "
                      "%.1f s of CPU work" % (fx, fy, spp, len(t), dt)}, frame


# MFMA roofline of the material MLP: the instruction the kernel issues decides the peak it is priced against (MI355X_MICROARCH.md)
MLP_ROOF = {"bound": "mfma", "peak": 157.3, "instruction": "v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate: the reference's fp32 Linear layers as fmaf chains, bit for bit)",
            "note": "algorithmic flops (4 480 per point) / event-timed duration against the fp32 MFMA peak (157.3 TFLOP/s); the kernel issues 6 144 flop per point "
                    "(the 6-row output layer occupies a 32-row tile): issued_frac = that rate against the same peak"}


def main():
    args = parse()
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)          # (only differs from LOCAL_RANK when ranks are forced onto fewer GPUs for a dry run with gloo)
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MIRRES_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for single-GPU dry runs of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    import mirres_restir_nerf_mesh_amd as M
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, dist as MD
    from mirres_restir_nerf_mesh_amd._ops import get_ctx
    from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
    S = M.scene
    dev = torch.device("cuda", local)

    # ---- synthetic scene of BASELINE config 2's shape
    v, t = S.make_mesh(args.subdiv, 64 if args.subdiv >= 6 else 16)
    W = RR.restirbvhWorker(torch.from_numpy(v).to(dev), torch.from_numpy(t).to(dev))
    W.update_mesh(W.vrt, W.v_ind)
    mlp = None
    if not args.const_material:
        params, w0, w1, w2 = S.make_matnet_params(seed=0)
        mn, mx = S.material_min_max()
        mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).to(dev), torch.from_numpy(mx).to(dev)))
        with torch.no_grad():
            mlp.encoder.params.copy_(torch.from_numpy(params).to(dev))
            for i, w in zip((0, 2, 4), (w0, w1, w2)):
                mlp.net.net[i].weight.copy_(torch.from_numpy(w).to(dev))
    g = harness.build_gbuffer(W, args.res, args.res, args.ssaa, mlp_mat=mlp)
    env = torch.from_numpy(S.make_env(256, 512)).to(dev)
    fx, fy = g["fx"], g["fy"]
    ctx = get_ctx(fx, fy, max_bounce=args.bounces)
    ctx.reserve()                       # the batch pool is sized here, not inside the first frame
    N = fx * fy

    def step(scheme):
        W.update_mesh(W.vrt, W.v_ind)                                   # LBVH rebuilt every frame (nerf/renderer.py:975)
        if world > 1 and scheme == "strips":          # exact: row strips + per-sample halo exchange + all-gather of the raw sums (dist.py)
            return MD.render_strips(ctx, W, mlp, env, g, args.spp, 12345, rank, world, max_bounce=args.bounces)
        return MD.render_sharded(ctx, W, mlp, env, g, args.spp, 12345, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(scheme):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(args.warmup):
            step(scheme)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o = step(scheme)
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d, o

    samples = float(N) * args.spp * args.steps
    schemes = ["spp"] if world == 1 else (["spp", "strips"] if args.shard == "both" else [args.shard])
    results = {}
    for i, sc in enumerate(schemes):
        if i == 0:
            results[sc] = timed(sc)
            continue
        # the second scheme must not take the line of the first one with it: an error that every rank raises (the likely kind: an API the backend refuses)
        # is caught, agreed on by all ranks and reported in the sub-record
        err = None
        try:
            results[sc] = timed(sc)
        except Exception as e:      # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        bad = torch.tensor([1.0 if err else 0.0], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if float(bad.item()) > 0:
            results[sc] = (None, err or "failed on another rank")
    primary = schemes[0]
    dt, out = results[primary]
    value = samples / dt / 1e6

    # ---- rooflines, measured live with HIP events on the launch stream (rank 0): the dominant kernel (shadow-ray traversal), the ordered
    #      closest-hit traversal, and the material MLP's GEMM phase
    roof = None
    if rank == 0 and not args.no_roofline:
        b, e = MD.spp_slice(args.spp, rank, world)
        prof_spp = max(1, min(8, e - b))
        def frame(n):
            occ = g["occ"].clone()
            RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, occ, g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], n, 2, 2, 2.0, 0.1, 0.001, 12345)
        # (a) visit counts of exactly these rays (deterministic; untimed): the production kernels' own (64-byte records they fetch) and, for the
        #     reference-equivalent figure, the reference traversal's (bvh_hit's order, no early exit) on the same ray set
        ctx.set_instrument(5); ctx.stats(reset=True); frame(prof_spp); st = ctx.stats(reset=True)
        ctx.set_instrument(1); ctx.stats(reset=True); frame(prof_spp); own = ctx.stats(reset=True)
        # (b) event-timed launches of the production kernels on the same rays
        ctx.set_instrument(2); ctx.trace_time(); frame(prof_spp); ms_any, n_any, ms_cl, n_cl = ctx.trace_time()
        ctx.set_instrument(0)
        rays_any, rays_cl = st["rays_any"], st["rays_closest"]
        total_rays = rays_any + rays_cl
        # ALGORITHMIC bytes of the kernel as built: every 64-byte node / leaf record the traversal has to read for its answer
        # (counted by the instrumented kernel on these very rays) + the 32-byte ray + the result (4 B hit flag; 28 B hit record for the closest hit).
        # (the spatial pass's rays reach the kernel as 8-byte pixel pairs and are formed there from two 16-byte gathers: 40 B requested, accounted as 32 B like the others)
        own_bytes_any = 64.0 * own["entered"] + rays_any * (32 + 4)
        own_bytes_cl = 64.0 * own["cl_entered"] + rays_cl * (32 + 28)
        # SURVEY section 8d's accounting (reference node layout, reference traversal's visit counts): what the REFERENCE's bvh_hit would have to move
        # for the same rays — kept as `reference_equiv`, not as the roofline (the production kernel does not do that work: early exit, 4-wide tree)
        ref_bytes_any = 24.0 * st["popped"] + 24.0 * st["entered"] + 48.0 * st["leaves"] + rays_any * (24 + 12 + 4)
        sec_any, sec_cl = ms_any * 1e-3, ms_cl * 1e-3
        achieved = own_bytes_any / sec_any / 1e9 if sec_any > 0 else 0.0
        achieved_cl = own_bytes_cl / sec_cl / 1e9 if sec_cl > 0 else 0.0
        HBM, L2 = 8000.0, 34500.0       # GB/s: MI355X_MICROARCH.md (HBM3E spec peak; aggregate L2 of the eight XCDs)
        pmc = {}
        for fn_ in ("r03_pmc_any4q_summary.json", "pmc_traffic.json"):
            try:
                pmc.update(json.load(open(os.path.join(ROOT, "profiles", fn_))))
            except Exception:
                pass
        roof = {"bound": "hbm", "kernel": "k_trace_any4q (shadow-ray BVH traversal)", "achieved": round(achieved, 2), "peak": HBM, "unit": "GB/s",
                "frac": round(achieved / HBM, 5), "traffic": pmc.get("k_trace_any_hbm_bytes_per_launch"),
                "bytes_per_ray": round(own_bytes_any / max(1, rays_any), 1), "launch_ms": round(ms_any / max(1, n_any), 4), "launches": n_any,
                "rays_per_launch": round(rays_any / max(1, n_any)), "grays_per_s": round(rays_any / sec_any / 1e9, 3) if sec_any > 0 else 0.0,
                "traffic_over_own_bytes": (round(pmc["k_trace_any_hbm_bytes_per_launch"] / (own_bytes_any / max(1, n_any)), 3) if pmc.get("k_trace_any_hbm_bytes_per_launch") else None),
                "l2_peak": L2, "l2_frac": round(achieved / L2, 5),
                "binding": "valu-issue", "valu_busy": pmc.get("valu_busy"), "lane_util": pmc.get("lane_util"), "l1_hit": pmc.get("l1_hit"),
                "reference_equiv": {"bytes_per_ray": round(ref_bytes_any / max(1, rays_any), 1), "tbps": round(ref_bytes_any / sec_any / 1e12, 2) if sec_any > 0 else 0.0,
                                    "note": "SURVEY 8d accounting: reference node layout x the reference traversal's visit counts on the same rays / this kernel's time"},
                "note": "achieved = bytes the kernel's own algorithm has to move (64-B records visited + ray + result, counted on the timed rays) / event-timed duration, against the HBM "
                        "peak as the contract asks; the 43 MB layout is cache resident (traffic = HBM bytes per launch from PMC — the spatial pass's pixel records and reservoirs gathered at the refill, "
                        "ray queues, results — `traffic_over_own_bytes` of the accounted bytes; 88 % L1 hits), so the "
                        "binding resource is VALU issue, not bandwidth: valu_busy = SQ_ACTIVE_INST_VALU / SIMD cycles, lane_util = active lanes per issued VALU instruction "
                        "(profiles/r03_pmc_any4q_summary.json)",
                "closest": {"kernel": "k_trace_closest4 (+ reference-order redo)", "achieved": round(achieved_cl, 2), "peak": HBM, "unit": "GB/s", "frac": round(achieved_cl / HBM, 5),
                            "bytes_per_ray": round(own_bytes_cl / max(1, rays_cl), 1), "launch_ms": round(ms_cl / max(1, n_cl), 4), "launches": n_cl,
                            "grays_per_s": round(rays_cl / sec_cl / 1e9, 3) if sec_cl > 0 else 0.0},
                "rays_per_pixel_sample": round(total_rays / (float(N) * prof_spp), 3),
                "per_ray": {"any_reference": [round(st[k] / max(1, rays_any), 2) for k in ("popped", "entered", "leaves")], "any_production": [round(own[k] / max(1, rays_any), 2) for k in ("popped", "entered", "leaves")],
                            "closest": [round(own[k] / max(1, rays_cl), 2) for k in ("cl_popped", "cl_entered", "cl_leaves")]},
                "traversal_share_of_step": round((ms_any + ms_cl) / prof_spp * args.spp / (dt / args.steps * 1e3) * (world if world > 1 else 1), 3)}
        if mlp is not None:
            # material MLP, GEMM phase alone (mirres_matnet_mlp on precomputed encodings of N points): 4 480 flop per point
            gen = torch.Generator(device=dev).manual_seed(0)
            enc = mlp.encode(torch.rand((N, 3), device=dev, generator=gen) * 1.2 - 0.6)
            mlp.mlp_on_encoding(enc); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                mlp.mlp_on_encoding(enc)
            e1.record(); torch.cuda.synchronize()
            ms_mlp = e0.elapsed_time(e1) / 20
            tf = 4480.0 * N / (ms_mlp * 1e-3) / 1e12
            roof["mlp"] = dict(MLP_ROOF, kernel="k_mlp_mfma (material MLP, GEMM phase)", achieved=round(tf, 2), unit="TFLOP/s", frac=round(tf / MLP_ROOF["peak"], 5),
                               issued_frac=round(tf * 6144.0 / 4480.0 / MLP_ROOF["peak"], 5), launch_ms=round(ms_mlp, 4), points=N, mfma_busy=pmc.get("mlp_mfma_busy"))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cf = cpu_baseline(S, args)
        # the same small frame (same inputs, seed and sample count) through the HIP path: the PSNR half of BASELINE's metric, against the CPU oracle
        cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        small, _, _ = RR.render_fused(get_ctx(cf["fx"], cf["fy"], max_bounce=args.bounces), W, mlp, False, (1, 1, 1), cu(cf["env"]), cu(cf["occ"][:, None]), cu(cf["normal"]),
                                      cu(cf["depth"][:, None]), cu(cf["kd"]), cu(cf["rm"]), cu(cf["ray_dir"]), cu(cf["pos"]), cf["spp"], 2, 2, 2.0, 0.1, 0.001, 12345)
        got_small = small[0].detach().cpu().numpy(); want_small = cf["final_color"]
        err = np.abs(got_small - want_small).max(axis=1)
        mse = float(np.mean((np.clip(got_small, 0, 1) - np.clip(want_small, 0, 1)) ** 2))
        cpu["psnr_hip_vs_oracle_db"] = round(-10.0 * float(np.log10(max(mse, 1e-20))), 2)
        cpu["max_abs_err"] = float(err.max()); cpu["frac_within_1e-3"] = float((err <= 1e-3).mean()); cpu["frac_bit_equal"] = float((err == 0).mean())
        # the north-star's parity bar on the benched workload: every pixel of the small frame within 1e-3 per channel of the CPU oracle
        assert cpu["max_abs_err"] <= 1e-3, "bench: HIP frame differs from the CPU oracle by %.3e (> 1e-3)" % cpu["max_abs_err"]

    if rank == 0:
        fc = out[0]
        par = {"spp": "sample slices + one all-reduce of the six accumulators (statistically equivalent frame)",
               "strips": "row strips + per-sample reservoir halo exchange + all-gather of radiance rows (bit-identical to one GPU)"}
        line = {"metric": "Msamples/s (pixels x spp), ReSTIR-DI + %d-bounce path tracing forward render" % (args.bounces + 1),
                "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": "BASELINE metric frame (configs[1] geometry at the metric's 512 spp): TensoIR-lego-shaped synthetic mesh (T=%d), %dx%d output, ssaa %d (internal %dx%d), "
                                       "%d spp, %d indirect bounces (3-bounce paths) + ReSTIR (initial/temporal/spatial), LBVH rebuild per frame, %s, EAW denoise"
                                       % (len(t), args.res, args.res, args.ssaa, fx, fy, args.spp, args.bounces, "constant material" if args.const_material else "hash-grid+MLP material field"),
                           "internal_pixels": N, "spp": args.spp, "triangles": int(len(t)),
                           "parallelism": ("%s x%d" % (par[primary], world)) if world > 1 else "single GPU",
                           "output_pixel_msamples_per_s": round(args.res * args.res * args.spp * args.steps / dt / 1e6, 3),
                           "finite": bool(torch.isfinite(fc).all().item()), "mean_radiance": round(float(fc[g["occ"][:, 0] > 0.5].mean().item()), 5)},
                "roofline": roof, "cpu_baseline": cpu}
        for sc in schemes[1:]:            # the other sharding scheme, timed the same way in the same run
            d2, e2 = results[sc]
            if d2 is None:
                line[sc] = {"value": None, "error": str(e2)[:300], "parallelism": "%s x%d" % (par[sc], world)}
                continue
            line[sc] = {"value": round(samples / d2 / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(d2 / args.steps * 1e3, 2), "parallelism": "%s x%d" % (par[sc], world)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
