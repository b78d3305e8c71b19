#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json on MI355X: Msamples/s (pixels x spp) of the ReSTIR + path-tracing forward render.

    python bench.py [--gpus N --steps K --warmup W] [--res 800 --ssaa 2 --spp 512 --bounces 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one frame of the hot path: LBVH rebuild (restirbvhWorker.update_mesh, as render_stage1 does every frame) + run_restir_di_with_pt
(spp loop: light tiles, initial / temporal / spatial reservoir passes, final shading, 2 indirect bounces with the material network, EAW
denoise, composite) over a synthetic scene of BASELINE config 2's shape, with every input already resident in HBM. The G-buffer (primary
visibility, nvdiffrast's job in the reference) is built once outside the timed region.

N > 1: the one frame is shared by the ranks (mirres-restir_nerf_mesh_amd/dist.py) — total work is fixed, i.e. STRONG scaling, as BASELINE's "at
1/2/4/8 GPU" — and BOTH sharding schemes are timed in the same run: `strips` = the north-star's pixel split (row strips, per-sample reservoir halo exchange,
all-gather of radiance rows; bit-identical to one GPU; boundaries balanced from the strips' measured times) and `spp` = sample slices + one sum over the ranks of the
accumulators the finish reads (dist.sum_over_ranks: all-to-all of slices, rank-ordered local sum, all-gather over RCCL; statistically equivalent frame). `value` is the
DECLARED scheme's (--value-scheme, default `spp`; `config.value_scheme` repeats it); the other scheme is reported beside it with its ratio.

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
    p.add_argument("--mesh", choices=("icosphere", "clustered"), default="icosphere",
                   help="icosphere: SURVEY 8d's synthetic mesh (uniform tessellation); clustered: the lego-like assembly of scene.make_mesh_clustered (studs, cavities, "
                        "thin plates, triangle areas spread > 1e5 : 1). The default run reports the other mesh as a sub-record")
    p.add_argument("--no-extras", action="store_true", help="skip the sub-records of the default single-GPU run (the other mesh, the configs[2] training step)")
    p.add_argument("--shard", choices=("both", "strips", "spp"), default="both",
                   help="N > 1: `spp` = sample slices + one sum of the accumulators over the ranks, `strips` = exact row strips + per-sample halo exchange + all-gather (the north-star's "
                        "tile split), `both` = time the two schemes one after the other; `value` is --value-scheme's")
    p.add_argument("--value-scheme", choices=("spp", "strips"), default="spp",
                   help="N > 1 with both schemes timed: which one the line's `value` is — declared here, not chosen after the fact (ADVICE r5); the other is reported beside it")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--const-material", action="store_true", help="constant material instead of the hash-grid + MLP field")
    return p.parse_args()


def cpu_baseline(S, args):
    """The CPU oracle (restatement of the reference kernels — NOT reference code, which is CUDA-only) on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as O
    v, t = S.mesh_by_name(args.mesh, args.subdiv)
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


MESH_WORDS = {"icosphere": "lego-SIZED synthetic mesh (noise-displaced icosphere + ground, 335 872 uniformly tessellated triangles: the mesh SURVEY 8d prescribes; BASELINE's TensoIR-lego has "
                           "this triangle COUNT but not this uniformity)",
              "clustered": "lego-LIKE synthetic mesh (brick assembly: studs, cavities, thin plates, 331 274 triangles with areas spread > 1e5 : 1 — the closer of the two to BASELINE's TensoIR-lego "
                           "in what a ray meets: 100 vs 63 reference node visits per shadow ray, 63 % vs 22 % occluded)"}


# MFMA roofline of the material MLP: the instruction the kernel issues decides the peak it is priced against (MI355X_MICROARCH.md)
MLP_ROOF = {"bound": "mfma", "peak": 157.3, "instruction": "v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate: the reference's fp32 Linear layers as fmaf chains, bit for bit)",
            "note": "algorithmic flops (4 480 per point) / event-timed duration against the fp32 MFMA peak (157.3 TFLOP/s); the kernel issues 6 144 flop per point "
                    "(the 6-row output layer occupies a 32-row tile): issued_frac = that rate against the same peak"}


# Which sources a counter snapshot depends on (round 6): a change in raster.hip must not stale the traversal counters. Every group holds the shared
# headers; "traversal" = the shadow-ray / closest-hit kernels and the hierarchy they walk; "chain" = every kernel of a frame; "train" = + the backward kernels.
_HDRS = ("engine.hpp", "device_math.hpp", "device_brdf.hpp", "device_grid.hpp", "device_light.hpp")
SHA_GROUPS = {"traversal": ("bvh_trace.hip", "bvh_build.hip") + _HDRS,
              "chain": ("bvh_trace.hip", "bvh_build.hip", "passes.hip", "shading.hip", "matnet.hip", "render.hip") + _HDRS,
              "train": ("bvh_trace.hip", "bvh_build.hip", "passes.hip", "shading.hip", "matnet.hip", "render.hip", "backward.hip", "eaw.hip") + _HDRS}
SNAPSHOT_GROUP = {"pmc_any4q_summary.json": "traversal", "pmc_traffic.json": "chain", "pmc_chain_summary.json": "chain", "pmc_train_summary.json": "train"}


def csrc_sha(group=None):
    """Identity of the native sources a counter snapshot belongs to (the GPU box has no .git): sha256 over the files of `group` (SHA_GROUPS) + include/*.h; without a
    group, over every csrc/*.hip, csrc/*.hpp and include/*.h (the identity A/B files and run logs carry)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mirres-restir_nerf_mesh_amd", "csrc")
    names = sorted(f for f in os.listdir(d) if f.endswith((".hip", ".hpp"))) if group is None else sorted(SHA_GROUPS[group])
    files = [os.path.join(d, f) for f in names] + sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h"))
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def pmc_snapshot():
    """Hardware-counter figures cannot be collected inside this process (rocprofv3 --pmc runs are separate, scripts/pmc_*.sh); the committed summaries are
    reported ONLY while they belong to the sources the library was built from (the snapshot records csrc_sha at collection time). Otherwise: stale, no numbers."""
    snap = {}
    for fn_ in SNAPSHOT_GROUP:
        path = os.path.join(ROOT, "profiles", fn_)
        try:
            d = json.load(open(path))
        except Exception:
            continue
        d["_file"] = "profiles/" + fn_
        snap[fn_] = d
    out = {"csrc_sha_now": {g_: csrc_sha(g_) for g_ in sorted(set(SNAPSHOT_GROUP.values()))}}
    for fn_, d in snap.items():
        grp = SNAPSHOT_GROUP[fn_]
        fresh = d.get("csrc_sha") == out["csrc_sha_now"][grp] and d.get("csrc_sha_group") == grp      # (snapshots of rounds 1-5 carry the all-files hash and no group: stale)
        key = "counters" if "any4q" in fn_ else ("chain" if "chain" in fn_ else ("train" if "train" in fn_ else "traffic"))
        if fresh:
            out[key] = dict({k: v for k, v in d.items() if not k.startswith("_")}, source="static:%s@%s" % (d["_file"], d.get("csrc_sha")))
        else:
            out[key] = {"stale": True, "source": "static:%s@%s" % (d["_file"], d.get("csrc_sha")), "note": "collected on other kernel sources: not reported"}
    return out


def make_field(S, torch, dev):
    from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D
    params, w0, w1, w2 = S.make_matnet_params(seed=0)
    mn, mx = S.material_min_max()
    mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).to(dev), torch.from_numpy(mx).to(dev)))
    with torch.no_grad():
        mlp.encoder.params.copy_(torch.from_numpy(params).to(dev))
        for i, w in zip((0, 2, 4), (w0, w1, w2)):
            mlp.net.net[i].weight.copy_(torch.from_numpy(w).to(dev))
    return mlp


def traversal_roofline(args, ctx, W, mlp, env, g, prof_spp, step_ms, world, pmc):
    """Live measurement (HIP events on the launch stream + the instrumented kernels' visit counts on the same rays) of the dominant kernel, k_trace_any4q."""
    import torch
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR
    N = g["fx"] * g["fy"]
    if float(N) * args.spp >= W.SAH_TOP_FROM_PIXEL_SAMPLES:
        W.upgrade()      # the short profiling frames below traverse the hierarchy the benched frame does (round 6: the SAH top is added for long frames only)
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
    cnt = (pmc or {}).get("counters") or {}; trf = (pmc or {}).get("traffic") or {}
    # the snapshot holds one counter set per mesh (collected on that mesh's rays): report the one that belongs to THIS run's mesh
    cnt_mesh = cnt if args.mesh == "icosphere" else (cnt.get(args.mesh) or {})
    valu_busy = None if cnt.get("stale") else cnt_mesh.get("valu_busy")
    lane_util = None if cnt.get("stale") else cnt_mesh.get("lane_util")
    trf_mesh = trf if args.mesh == "icosphere" else (trf.get(args.mesh) or {})
    traffic = None if trf.get("stale") else trf_mesh.get("k_trace_any_hbm_bytes_per_launch")
    chain = (pmc or {}).get("chain") or {}
    chain_mesh = None if (chain.get("stale") or not chain) else (chain.get(args.mesh) or None)
    launch_s = sec_any / max(1, n_any)
    # which resource binds the kernel is read off the counters, not asserted: VALU pipes busy for most of the SIMD cycles while the HBM counters show a fraction of
    # the peak -> "valu-issue"; without a counter snapshot that belongs to the built sources the record says so
    hbm_frac = (traffic / launch_s / 1e9 / HBM) if (traffic and launch_s > 0) else None
    valu_useful = round(valu_busy * lane_util, 4) if (valu_busy is not None and lane_util is not None) else None
    if valu_busy is None:
        bound = "unknown (no counter snapshot for these kernel sources: run scripts/profile_r04.sh)"
    elif valu_busy >= 0.6 and (hbm_frac is None or hbm_frac < valu_busy):
        bound = "valu-issue"
    else:
        bound = "hbm"
    roof = {"bound": bound, "kernel": "k_trace_any4q (shadow-ray BVH traversal)",
            # the binding resource is VALU issue (the 43 MB node / leaf layout is cache resident): frac = SQ_ACTIVE_INST_VALU / SIMD cycles of the kernel, a hardware-counter
            # figure from the snapshot below (null when the snapshot does not belong to the built sources); everything else in this record is measured in this run
            # frac is the THROUGHPUT fraction of the VALU roof (round 6): busy x lane utilisation = the share of the SIMDs' lane-cycles that do useful
            # lane work (SQ_THREAD_CYCLES_VALU / (64 lanes x SIMD cycles)); the busy fraction alone (the SIMDs issue a VALU instruction in that share of the kernel's
            # cycles — divergence between node and leaf lanes and idle lanes between refills are inside it) is kept beside it as valu_busy
            "achieved": (round(traffic / launch_s / 1e9, 1) if bound == "hbm" else valu_useful), "peak": (HBM if bound == "hbm" else 1.0),
            "unit": ("GB/s" if bound == "hbm" else "useful VALU lane-cycles per lane-cycle of the 1024 SIMDs"), "frac": (round(hbm_frac, 4) if bound == "hbm" else valu_useful),
            "valu_busy": valu_busy, "valu_useful": valu_useful, "lane_util": lane_util,
            "traffic": traffic,
            "own_bytes": {"achieved": round(achieved, 2), "unit": "GB/s", "bytes_per_ray": round(own_bytes_any / max(1, rays_any), 1),
                          "l2_peak": L2, "l2_frac": round(achieved / L2, 5), "hbm_peak": HBM, "over_hbm_peak": round(achieved / HBM, 5),
                          "note": "bytes the kernel's own algorithm requests from global memory (64-B records visited, counted on the timed rays, + ray + result) / event-timed duration. "
                                  "About 90 % of the requests are L1 hits: this is a request rate, NOT an HBM utilisation (it exceeds the HBM peak) — the DRAM side is hbm_counter"},
            "hbm_counter": ({"bytes_per_launch": traffic, "achieved": round(traffic / launch_s / 1e9, 1), "peak": HBM, "unit": "GB/s", "frac": round(traffic / launch_s / 1e9 / HBM, 4),
                             "over_own_bytes": round(traffic / (own_bytes_any / max(1, n_any)), 3),
                             "note": "FETCH_SIZE + WRITE_SIZE per launch (snapshot, collected on THIS mesh) / this run's launch time"} if traffic and launch_s > 0 else None),
            "launch_ms": round(ms_any / max(1, n_any), 4), "launches": n_any,
            "rays_per_launch": round(rays_any / max(1, n_any)), "grays_per_s": round(rays_any / sec_any / 1e9, 3) if sec_any > 0 else 0.0,
            "pmc_snapshot": pmc,
            "reference_equiv": {"bytes_per_ray": round(ref_bytes_any / max(1, rays_any), 1), "tbps": round(ref_bytes_any / sec_any / 1e12, 2) if sec_any > 0 else 0.0,
                                "note": "SURVEY 8d accounting: reference node layout x the reference traversal's visit counts on the same rays / this kernel's time"},
            "closest": {"kernel": "k_trace_closest4 (+ reference-order redo)", "own_bytes_gbps": round(achieved_cl, 2), "own_bytes_l2_frac": round(achieved_cl / L2, 5),
                        "bytes_per_ray": round(own_bytes_cl / max(1, rays_cl), 1), "launch_ms": round(ms_cl / max(1, n_cl), 4), "launches": n_cl,
                        "grays_per_s": round(rays_cl / sec_cl / 1e9, 3) if sec_cl > 0 else 0.0, "redo_frac": round(own["cl_redo"] / max(1, rays_cl), 6)},
            "rays_per_pixel_sample": round(total_rays / (float(N) * prof_spp), 3),
            "per_ray": {"any_reference": [round(st[k] / max(1, rays_any), 2) for k in ("popped", "entered", "leaves")], "any_production": [round(own[k] / max(1, rays_any), 2) for k in ("popped", "entered", "leaves")],
                        "closest": [round(own[k] / max(1, rays_cl), 2) for k in ("cl_popped", "cl_entered", "cl_leaves")],
                        "legend": "reference: nodes popped / internal nodes entered / leaves tested by bvh_hit's own order; production: box tests / 64-B records fetched / leaves tested"},
            "private_stack_deepest": {"shadow": own["any_max_stack"], "ordered_closest": own["cl_max_stack"], "overflows": own["any_stack_overflow"]},
            # queue entries of the spatial pass that are answered without a traversal (light reservoir with luminance 0: the merge cannot see the answer; the reference
            # traces them). They ARE counted in rays_per_launch / grays_per_s / per_ray (as the one root-box test they cost); MIRRES_SKIP_DEAD=0 traces them
            "rays_not_traced_frac": round(own.get("any_dead", 0) / max(1, rays_any), 4),
            "traversal_share_of_step": round((ms_any + ms_cl) / prof_spp * args.spp / step_ms * (world if world > 1 else 1), 3),
            # the rest of the per-sample chain (k_spatial_gen, k_spatial_resolve) and the batched candidate loop (k_initial_gen): hardware counters of one serialised
            # frame on this mesh (scripts/pmc_chain.sh; snapshot, csrc_sha-gated like the others) — which resource each of them is short of
            "chain": chain_mesh}
    # the STEP against the VALU-issue roof (round 5): wave64 VALU instructions of one sample (counter snapshot of a serialised frame on this mesh) / what 1024 SIMDs can
    # issue in this run's time per sample at one instruction per 4 cycles and the 2.4 GHz peak clock — the frame's own roofline fraction; a snapshot figure over a live time
    smp = (chain_mesh or {}).get("_sample") if chain_mesh else None
    if smp and step_ms > 0:
        per_sample_s = step_ms * 1e-3 / args.spp * (world if world > 1 else 1)
        roof["frame_valu_issue"] = {"wave_insts_per_sample": smp["valu_wave_insts_per_sample"], "peak_wave_insts_per_s": 1024 * 2.4e9 / 4,
                                    "frac": round(smp["valu_wave_insts_per_sample"] / (1024 * 2.4e9 / 4) / per_sample_s, 4),
                                    "note": "whole-frame VALU-issue utilisation: (VALU wave-instructions per sample, PMC snapshot) / (614.4 G per second x this run's time per sample)"}
    return roof


ATOMIC_ROOF_G = 21.0      # G scattered fp32 atomic REQUESTS per second the memory side of an MI355X takes (scripts/ubench/atomic_rate.hip, profiles/r06_atomic_rate.txt: the same for
                          # every scope — each device-scope atomic leaves the XCD's L2 —; the lanes of one instruction that fall into one 32-byte sector are one request)


def train_roofline(pmc, fg_points, spp):
    """configs[2]'s backward kernels against the roof that binds them (round 6): both scatter gradients with fp32 atomics — the hash-grid table (k_matnet_bwd: 16 levels x
    8 corners x 2 features = 256 adds per encoded foreground point) and the environment map (k_direct_bwd: 4 texels x 3 channels = 12 adds per visible final sample) —
    and run at the memory side's atomic REQUEST rate. Counter snapshot (scripts/pmc_train.sh; csrc_sha-gated like the others) over the measured roof."""
    tr = (pmc or {}).get("train") or {}
    if not tr or tr.get("stale"):
        return {"bound": "atomic-requests", "peak": ATOMIC_ROOF_G, "unit": "G requests/s", "stale_or_missing_snapshot": True, "source": tr.get("source")}
    out = {"bound": "atomic-requests", "peak": ATOMIC_ROOF_G, "unit": "G atomic requests/s (32-byte sectors) at the memory side", "kernels": {}, "source": tr.get("source"),
           "algorithmic": {"k_matnet_bwd": "256 fp32 adds per foreground point (%d points): %.1f M adds, two per 8-byte table entry -> %.1f M requests when a lane pair shares an entry" % (fg_points, 256e-6 * fg_points, 128e-6 * fg_points),
                           "k_direct_bwd": "12 fp32 adds per visible final sample (<= %d x %d samples): <= %.1f M adds, three per 12-byte texel -> a third of the requests when four lanes share a texel" % (fg_points, spp, 12e-6 * fg_points * spp)}}
    for k in ("k_direct_bwd", "k_matnet_bwd"):
        c = tr.get(k)
        if not c:
            continue
        req, us = c.get("atomic_requests_per_launch"), c.get("launch_us")
        out["kernels"][k] = dict(c, achieved=(round(req / us / 1e3, 2) if req and us else None), frac=(round(req / us / 1e3 / ATOMIC_ROOF_G, 4) if req and us else None))
    ks = out["kernels"]
    if ks:
        req = sum(v.get("atomic_requests_per_launch") or 0 for v in ks.values()); us = sum(v.get("launch_us") or 0 for v in ks.values())
        out["achieved"] = round(req / us / 1e3, 2) if us else None
        out["frac"] = round(req / us / 1e3 / ATOMIC_ROOF_G, 4) if us else None
    return out


def train_step_record(S, torch, dev, W, steps=3, pmc=None):
    """BASELINE configs[2]: one stage-1 inverse-rendering step — forward + backward through FinalShading / EvaluateFinalSamples_di / EAW / the material field,
    Adam step on field + environment — at 800 x 800, 32 spp (main.py:108), LBVH rebuild per step, on the benched mesh."""
    from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness
    mlp = make_field(S, torch, dev)
    g = harness.build_gbuffer(W, 800, 800, 1)
    fx, fy = g["fx"], g["fy"]; N = fx * fy
    mods = RR.load_m_for_restir(fx, fy)
    env = torch.full((256, 512, 3), 0.5, device=dev, requires_grad=True)
    opt = torch.optim.Adam([{"params": mlp.parameters(), "lr": 1e-3}, {"params": [env], "lr": 1e-2}])
    target = torch.rand((N, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 0.5 + 0.25
    fg = g["occ"][:, 0] > 0.5
    z = lambda *s_: torch.zeros(s_, device=dev)
    def step():
        opt.zero_grad(set_to_none=True)
        W.update_mesh(W.vrt, W.v_ind)
        kdks = mlp.sample(g["pos"])
        kd = kdks[:, 0:3].contiguous(); rm = torch.cat((kdks[:, 4:5], kdks[:, 5:6]), -1).contiguous()
        out = RR.run_restir_di_with_pt(False, 1.0, 1.0, 1.0, mlp, None, W, *mods[:8], *mods[8:17], env, g["occ"].clone(), g["normal"], g["depth"], kd, rm, g["ray_dir"], g["pos"],
                                       z(N, 1), z(N, 4), z(N, 3), z(N, 3), fx, fy, 32, 2, 2, 2.0, 0.1, 0.001)
        loss = (torch.clamp(out[0][fg], 0, 1) - target[fg]).abs().mean()
        loss.backward(); opt.step()
        with torch.no_grad():
            env.clamp_(min=0.01)
        return loss
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"config": "BASELINE configs[2]: stage-1 training step, 800x800, 32 spp, forward + backward (direct-lighting terms) + Adam, LBVH rebuild per step",
            "ms_per_step": round(dt * 1e3, 2), "steps": steps, "warmup": 1, "msamples_per_s": round(N * 32 / dt / 1e6, 1), "loss_finite": bool(torch.isfinite(last).item()),
            "roofline": train_roofline(pmc, int(fg.sum().item()), 32)}


def main():
    args = parse()
    # stdout carries the ONE JSON line and nothing else: whatever the mirrored reference modules print on the way (load_m_for_restir's "Create neighbor offset time
    # consumed", as nerf/renderer_restir.py:227 does) goes to stderr
    out_stream = sys.stdout
    sys.stdout = sys.stderr
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
    S = M.scene
    dev = torch.device("cuda", local)

    # ---- synthetic scene of BASELINE config 2's size
    def scene_for(mesh_name):
        v, t = S.mesh_by_name(mesh_name, args.subdiv)
        W = RR.restirbvhWorker(torch.from_numpy(v).to(dev), torch.from_numpy(t).to(dev))
        W.update_mesh(W.vrt, W.v_ind)
        mlp = None if args.const_material else make_field(S, torch, dev)
        g = harness.build_gbuffer(W, args.res, args.res, args.ssaa, mlp_mat=mlp)
        return t, W, mlp, g
    t, W, mlp, g = scene_for(args.mesh)
    env = torch.from_numpy(S.make_env(256, 512)).to(dev)
    fx, fy = g["fx"], g["fy"]
    ctx = get_ctx(fx, fy, max_bounce=args.bounces)
    if world == 1:
        ctx.reserve()                   # the batch pool is sized here, not inside the first frame
    # (N > 1: a rank renders spp / N samples, or a strip, in batches the library sizes for them — the warm-up frames carve those pools; reserving the default 64-sample
    # pool of the whole frame first would hold 113 GB the run never uses, which two dry-run ranks sharing one GPU do not have: scripts/runs/r06_run43.sh)
    N = fx * fy

    balancer = MD.StripBalancer(fy, world) if world > 1 else None      # strip boundaries follow the strips' measured times from frame to frame (bit-identical for any partition)

    def make_step(W_, mlp_, g_):
        def step(scheme):
            W_.update_mesh(W_.vrt, W_.v_ind)                                   # LBVH rebuilt every frame (nerf/renderer.py:975)
            if world > 1 and scheme == "strips":          # exact: row strips + per-sample halo exchange + all-gather of the raw sums (dist.py)
                return MD.render_strips(ctx, W_, mlp_, env, g_, args.spp, 12345, rank, world, max_bounce=args.bounces, balancer=balancer)
            return MD.render_sharded(ctx, W_, mlp_, env, g_, args.spp, 12345, rank, world)
        return step
    step = make_step(W, mlp, g)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(scheme, step_fn=None, warmup=None, steps=None):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
        step_fn = step_fn or step
        for _ in range(args.warmup if warmup is None else warmup):
            step_fn(scheme)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps if steps is None else steps):
            o = step_fn(scheme)
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
        # is caught, agreed on by all ranks and reported in the sub-record. A HANG would (the exact scheme's per-sample point-to-point exchange has never run on more
        # than one GPU: SCALE was skipped in every round): a watchdog prints the line of the first scheme — the declared value — with the second one marked as
        # timed out, and ends the process, if the second scheme takes more than 30 x the first one's time (at least three minutes).
        import threading
        d0 = results[schemes[0]][0]
        bail_note = ["timed out (watchdog): the scheme did not finish; the line carries the first scheme only"]
        def _bail():
            if rank == 0:
                line = {"metric": "Msamples/s (pixels x spp), ReSTIR-DI + %d-bounce path tracing forward render" % (args.bounces + 1),
                        "value": round(samples / d0 / 1e6, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                        "ms_per_step": round(d0 / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                        "config": {"workload": "BASELINE metric frame, %s mesh, %dx%d output, ssaa %d, %d spp" % (args.mesh, args.res, args.res, args.ssaa, args.spp), "value_scheme": schemes[0],
                                   "bit_identical_to_one_gpu": schemes[0] == "strips"},
                        "roofline": None, "cpu_baseline": None, sc: {"value": None, "error": bail_note[0]}}
                print(json.dumps(line), file=out_stream, flush=True)
            os._exit(0 if rank == 0 else 3)
        limit = max(180.0, 30.0 * d0 * (args.steps + args.warmup) / max(1, args.steps))
        if os.environ.get("MIRRES_BENCH_WATCHDOG_S"):      # (tests of the watchdog itself)
            limit = float(os.environ["MIRRES_BENCH_WATCHDOG_S"])
        dog = threading.Timer(limit, _bail)
        dog.daemon = True; dog.start()
        err = None
        try:
            results[sc] = timed(sc)
        except Exception as e:      # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
            bail_note[0] = "failed on rank %d (%s); the other ranks did not return from it" % (rank, err[:200])
            print("[bench] rank %d: scheme %s failed: %s" % (rank, sc, err), file=sys.stderr, flush=True)
        # the watchdog stays armed across the agreement: a rank that failed ALONE leaves the others inside the scheme (a per-sample exchange waits for it), and
        # this all-reduce would wait for them for ever
        bad = torch.tensor([1.0 if err else 0.0], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        dog.cancel()
        if float(bad.item()) > 0:
            results[sc] = (None, err or "failed on another rank")
    # Which scheme the line's `value` is: DECLARED (--value-scheme, default the sample slices), never chosen by the outcome — a consumer comparing rounds reads the same
    # quantity every time. Both schemes are timed the same way (W warm-up + K steps between barriers, max over ranks) and the other one (row strips: the north-star's
    # pixel split, bit-identical to one GPU) is reported beside it with its ratio. A scheme that failed on some rank cannot be the value.
    primary = schemes[0]
    if len(schemes) > 1 and results[schemes[1]][0] is not None and args.value_scheme in results and results[args.value_scheme][0] is not None:
        primary = args.value_scheme
    dt, out = results[primary]
    value = samples / dt / 1e6
    step_ms = dt / args.steps * 1e3

    # ---- rooflines, measured live with HIP events on the launch stream (rank 0): the dominant kernel (shadow-ray traversal), the ordered
    #      closest-hit traversal, and the material MLP's GEMM phase
    roof = None
    pmc = pmc_snapshot() if rank == 0 else None
    if rank == 0 and not args.no_roofline:
        b, e = MD.spp_slice(args.spp, rank, world)
        prof_spp = max(1, min(8, e - b))
        roof = traversal_roofline(args, ctx, W, mlp, env, g, prof_spp, step_ms, world, pmc)
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
            cnt = (pmc or {}).get("counters") or {}
            roof["mlp"] = dict(MLP_ROOF, kernel="k_mlp_mfma (material MLP, GEMM phase)", achieved=round(tf, 2), unit="TFLOP/s", frac=round(tf / MLP_ROOF["peak"], 5),
                               issued_frac=round(tf * 6144.0 / 4480.0 / MLP_ROOF["peak"], 5), launch_ms=round(ms_mlp, 4), points=N,
                               mfma_busy=None if cnt.get("stale") else cnt.get("mlp_mfma_busy"))

    # ---- sub-records of the default single-GPU run: the other mesh (same frame, same counters), the configs[2] training step
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        other = "clustered" if args.mesh == "icosphere" else "icosphere"
        try:
            t2, W2, mlp2, g2 = scene_for(other)
            d2, o2 = timed("spp", make_step(W2, mlp2, g2))          # the same --steps / --warmup as the headline: a first-class number, not a one-frame aside
            d2s = d2 / args.steps
            fg2 = float((g2["occ"][:, 0] > 0.5).float().mean().item())
            rec = {"mesh": other, "workload": MESH_WORDS[other], "triangles": int(len(t2)), "value": round(float(N) * args.spp / d2s / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(d2s * 1e3, 2),
                   "steps": args.steps, "warmup": args.warmup,
                   "foreground_frac": round(fg2, 4), "foreground_msamples_per_s": round(float(N) * args.spp * fg2 / d2s / 1e6, 3), "finite": bool(torch.isfinite(o2[0]).all().item())}
            if not args.no_roofline:
                args_o = argparse.Namespace(**vars(args)); args_o.mesh = other
                r2 = traversal_roofline(args_o, ctx, W2, mlp2, env, g2, 8 if args.spp >= 8 else args.spp, d2s * 1e3, 1, pmc)
                r2.pop("pmc_snapshot", None)                          # (printed once, in the headline's record)
                rec["roofline"] = r2
            extras[other] = rec
            del W2, mlp2, g2
        except Exception as e_:      # noqa: BLE001 — a sub-record must not take the line with it
            extras[other] = {"error": "%s: %s" % (type(e_).__name__, str(e_)[:300])}
        try:
            extras["train_step"] = train_step_record(S, torch, dev, W, pmc=pmc)
        except Exception as e_:      # noqa: BLE001
            extras["train_step"] = {"error": "%s: %s" % (type(e_).__name__, str(e_)[:300])}

    cpu = None
    parity_failed = None
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
        # the north-star's parity bar on the benched workload: every pixel of the small frame within 1e-3 per channel of the CPU oracle. A failure is
        # reported IN the line (and by the exit code), not instead of it
        if not cpu["max_abs_err"] <= 1e-3:
            parity_failed = "HIP frame differs from the CPU oracle by %.3e (> 1e-3) on the %dx%d x %d spp sample" % (cpu["max_abs_err"], cf["fx"], cf["fy"], cf["spp"])

    if rank == 0:
        fc = out[0]
        fgm = g["occ"][:, 0] > 0.5
        fg_frac = float(fgm.float().mean().item())
        par = {"spp": "sample slices + one sum over the ranks of the four accumulators the finish reads, foreground pixels only (all-to-all of slices, rank-ordered local sum, all-gather; statistically equivalent frame)",
               "strips": "row strips + per-sample reservoir halo exchange + all-gather of radiance rows (bit-identical to one GPU)"}
        mesh_words = MESH_WORDS[args.mesh]
        line = {"metric": "Msamples/s (pixels x spp), ReSTIR-DI + %d-bounce path tracing forward render" % (args.bounces + 1),
                "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(step_ms, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": "BASELINE metric frame (configs[1] geometry at the metric's 512 spp): %s, T=%d, %dx%d output, ssaa %d (internal %dx%d), "
                                       "%d spp, %d indirect bounces (3-bounce paths) + ReSTIR (initial/temporal/spatial), LBVH rebuild per frame, %s, EAW denoise"
                                       % (mesh_words, len(t), args.res, args.res, args.ssaa, fx, fy, args.spp, args.bounces, "constant material" if args.const_material else "hash-grid+MLP material field"),
                           "mesh": args.mesh, "internal_pixels": N, "spp": args.spp, "triangles": int(len(t)),
                           "parallelism": ("%s x%d" % (par[primary], world)) if world > 1 else "single GPU",
                           "output_pixel_msamples_per_s": round(args.res * args.res * args.spp * args.steps / dt / 1e6, 3),
                           # the metric counts every pixel of the frame; background pixels cost almost nothing, so the number scales with coverage:
                           "foreground_frac": round(fg_frac, 4), "foreground_msamples_per_s": round(value * fg_frac, 3),
                           "finite": bool(torch.isfinite(fc).all().item()), "mean_radiance": round(float(fc[fgm].mean().item()), 5)},
                "roofline": roof, "cpu_baseline": cpu}
        line.update(extras)
        if parity_failed:
            line["parity_failed"] = parity_failed
        for sc in schemes:                # the other sharding scheme, timed the same way in the same run
            if sc == primary:
                continue
            d2, e2 = results[sc]
            if d2 is None:
                line[sc] = {"value": None, "error": str(e2)[:300], "parallelism": "%s x%d" % (par[sc], world)}
                continue
            line[sc] = {"value": round(samples / d2 / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(d2 / args.steps * 1e3, 2), "parallelism": "%s x%d" % (par[sc], world),
                        "bit_identical_to_one_gpu": sc == "strips", "ratio_to_value": round(dt / d2, 4)}
        if world > 1:
            line["config"]["value_scheme"] = primary
            line["config"]["bit_identical_to_one_gpu"] = primary == "strips"
            if balancer is not None and balancer.history:
                b_, t_ = balancer.history[-1]
                line["config"]["strip_balance"] = {"bounds": b_, "strip_ms": [round(x, 2) for x in t_], "max_over_mean": round(max(t_) / (sum(t_) / len(t_)), 4),
                                                   "note": "last exchanged strip render times (dist.StripBalancer): boundaries follow them from frame to frame"}
                if balancer.last_wait_ms is not None and balancer.last_spp:      # rank 0's stream time inside the per-sample halo exchanges of its last timed strip render
                    line["config"]["strip_exchange"] = {"path": balancer.last_path, "rank0_frame_ms": round(balancer.last_total_ms, 2), "rank0_in_exchange_ms": round(balancer.last_wait_ms, 2),
                                                        "exchange_us_per_sample": round(balancer.last_wait_ms * 1e3 / balancer.last_spp, 1),
                                                        "note": "device time between events around every (spp / 32)-th exchange, scaled; includes waiting for the neighbour strips; "
                                                                "host cost per exchange: profiles/r06_halo_host_cost.txt"}
        print(json.dumps(line), file=out_stream, flush=True)
    if world > 1:
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench: " + parity_failed)


if __name__ == "__main__":
    main()
