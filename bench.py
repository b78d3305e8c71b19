#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json on MI355X: Msamples/s (pixels x spp) of the ReSTIR + path-tracing forward render.

    python bench.py [--gpus N --steps K --warmup W] [--res 800 --ssaa 2 --spp 128 --bounces 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one frame of the hot path: LBVH rebuild (restirbvhWorker.update_mesh, as render_stage1 does every frame) + run_restir_di_with_pt
(spp loop: light tiles, initial / temporal / spatial reservoir passes, final shading, 2 indirect bounces with the material network, EAW
denoise, composite) over a synthetic scene of BASELINE config 2's shape, with every input already resident in HBM. The G-buffer (primary
visibility, nvdiffrast's job in the reference) is built once outside the timed region.

N > 1: the frame's spp range is split across ranks (mirres-restir_nerf_mesh_amd/dist.py), one all-reduce (RCCL) of the accumulators, then
the replicated finish — total work is fixed, i.e. STRONG scaling of the one frame, as BASELINE's "at 1/2/4/8 GPU".

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
    p.add_argument("--spp", type=int, default=128)
    p.add_argument("--bounces", type=int, default=2, help="indirect bounces (MAX_Bounce, FinalShading.slang:7) -> 3 path vertices")
    p.add_argument("--subdiv", type=int, default=7, help="icosphere subdivisions (7 -> 327 680 + 8 192 ground triangles)")
    p.add_argument("--shard", choices=("strips", "spp"), default="spp", help="N > 1: exact row strips with halo exchange, or spp slices + all-reduce")
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
    N = fx * fy

    def step():
        W.update_mesh(W.vrt, W.v_ind)                                   # LBVH rebuilt every frame (nerf/renderer.py:975)
        if world > 1 and args.shard == "strips":      # exact: row strips + per-sample halo exchange + all-gather of the raw sums (dist.py)
            return MD.render_strips(ctx, W, mlp, env, g, args.spp, 12345, rank, world, max_bounce=args.bounces)
        return MD.render_sharded(ctx, W, mlp, env, g, args.spp, 12345, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    samples = float(N) * args.spp * args.steps
    value = samples / dt / 1e6

    # ---- roofline of the dominant kernel (any-hit traversal), measured live with HIP events on the launch stream (rank 0)
    roof = None
    if rank == 0 and not args.no_roofline:
        b, e = MD.spp_slice(args.spp, rank, world)
        prof_spp = max(1, min(8, e - b))
        def frame(n):
            occ = g["occ"].clone()
            RR.render_fused(ctx, W, mlp, False, (1, 1, 1), env, occ, g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], n, 2, 2, 2.0, 0.1, 0.001, 12345)
        # (a) visit counts of exactly these rays (deterministic; untimed): the reference traversal's own counts ("the reference node
        #     layout as the accounting basis regardless of the build's internal layout", counts of bvh_hit's order on the same ray set) and the
        #     production kernel's (64-byte records it fetches from global memory)
        ctx.set_instrument(5); ctx.stats(reset=True); frame(prof_spp); st = ctx.stats(reset=True)
        ctx.set_instrument(1); ctx.stats(reset=True); frame(prof_spp); own = ctx.stats(reset=True)
        # (b) event-timed launches of the production kernels on the same rays
        ctx.set_instrument(2); ctx.trace_time(); frame(prof_spp); ms_any, n_any, ms_cl, n_cl = ctx.trace_time()
        ctx.set_instrument(0)
        rays_any, rays_cl = st["rays_any"], st["rays_closest"]
        # algorithmic bytes: per ray 24 (o,d) + 12 (root info) + result, 24 per popped node (aabb) + 24 per entered internal node
        # (two child infos) + 48 per tested leaf (indices + vertices), with the reference traversal's counts.
        total_rays = rays_any + rays_cl
        bytes_any = 24.0 * st["popped"] + 24.0 * st["entered"] + 48.0 * st["leaves"] + rays_any * (24 + 12 + 4)
        bytes_cl = 24.0 * st["cl_popped"] + 24.0 * st["cl_entered"] + 48.0 * st["cl_leaves"] + rays_cl * (24 + 12 + 28)
        own_bytes_any = 64.0 * own["entered"] + rays_any * (32 + 4)      # compressed 64-byte node / leaf records + the 32-byte ray + result
        ms_launch = ms_any / max(1, n_any)
        achieved = bytes_any / (ms_any * 1e-3) / 1e9 if ms_any > 0 else 0.0
        own_achieved = own_bytes_any / (ms_any * 1e-3) / 1e9 if ms_any > 0 else 0.0
        roof = {"bound": "hbm", "kernel": "k_trace_any4q (shadow-ray BVH traversal)", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": None,
                "launch_ms": round(ms_launch, 4), "launches": n_any, "rays_per_launch": round(rays_any / max(1, n_any)),
                "bytes_per_ray": round(bytes_any / max(1, rays_any), 1), "grays_per_s": round(rays_any / (ms_any * 1e-3) / 1e9, 3) if ms_any > 0 else 0.0,
                "own_bytes_per_ray": round(own_bytes_any / max(1, rays_any), 1), "own_achieved": round(own_achieved, 2), "own_frac": round(own_achieved / 8000.0, 5),
                "note": "achieved = SURVEY 8d algorithmic bytes (reference layout, reference traversal's visit counts on the same rays) / event-timed duration; "
                        "it exceeds the HBM peak because the production kernel stops at the first occluder, walks a 4-wide tree near-first and fetches 64-byte compressed "
                        "records (own_*: the bytes it actually requests), and the 43 MB layout is cache resident (traffic = HBM bytes from PMC); the kernel is bound by issue slots and the "
                        "dependent chain of a traversal step, not by memory: fetching every record twice costs +1.5 % (DESIGN.md section 5, round 2)",
                "closest_launch_ms": round(ms_cl / max(1, n_cl), 4), "closest_achieved": round(bytes_cl / (ms_cl * 1e-3) / 1e9, 2) if ms_cl > 0 else 0.0, "rays_per_pixel_sample": round(total_rays / (float(N) * prof_spp), 3),
                "per_ray": {"any_reference": [round(st[k] / max(1, rays_any), 2) for k in ("popped", "entered", "leaves")], "any_production": [round(own[k] / max(1, rays_any), 2) for k in ("popped", "entered", "leaves")],
                            "closest": [round(st[k] / max(1, rays_cl), 2) for k in ("cl_popped", "cl_entered", "cl_leaves")]},
                "traversal_share_of_step": round((ms_any + ms_cl) / prof_spp * args.spp / (dt / args.steps * 1e3) * (world if world > 1 else 1), 3)}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                roof["traffic"] = json.load(open(pmc)).get("k_trace_any_hbm_bytes_per_launch")
            except Exception:
                pass
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cf = cpu_baseline(S, args)
        # the same small frame (same inputs, seed and sample count) through the HIP path: the PSNR half of BASELINE's metric, against the CPU oracle
        cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        small, _, _ = RR.render_fused(get_ctx(cf["fx"], cf["fy"], max_bounce=args.bounces), W, mlp, False, (1, 1, 1), cu(cf["env"]), cu(cf["occ"][:, None]), cu(cf["normal"]),
                                      cu(cf["depth"][:, None]), cu(cf["kd"]), cu(cf["rm"]), cu(cf["ray_dir"]), cu(cf["pos"]), cf["spp"], 2, 2, 2.0, 0.1, 0.001, 12345)
        mse = float(((small[0].clamp(0, 1) - cu(cf["final_color"]).clamp(0, 1)) ** 2).mean().item())
        cpu["psnr_hip_vs_oracle_db"] = round(-10.0 * float(np.log10(max(mse, 1e-20))), 2)

    if rank == 0:
        fc = out[0]
        line = {"metric": "Msamples/s (pixels x spp), ReSTIR-DI + %d-bounce path tracing forward render" % (args.bounces + 1),
                "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": "BASELINE configs[1]: TensoIR-lego-shaped synthetic mesh (T=%d), %dx%d output, ssaa %d (internal %dx%d), %d spp, "
                                       "%d indirect bounces + ReSTIR (initial/temporal/spatial), LBVH rebuild per frame, %s, EAW denoise"
                                       % (len(t), args.res, args.res, args.ssaa, fx, fy, args.spp, args.bounces, "constant material" if args.const_material else "hash-grid+MLP material field"),
                           "internal_pixels": N, "spp": args.spp, "triangles": int(len(t)), "parallelism": ("%s x%d" % ("row strips + halo exchange + all-gather" if args.shard == "strips" else "spp-sharded + all-reduce", world)) if world > 1 else "single GPU",
                           "output_pixel_msamples_per_s": round(args.res * args.res * args.spp * args.steps / dt / 1e6, 3),
                           "finite": bool(torch.isfinite(fc).all().item()), "mean_radiance": round(float(fc[g["occ"][:, 0] > 0.5].mean().item()), 5)},
                "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
