"""Stage-1 inverse rendering on the HIP path (the loop of Trainer.train_one_epoch, nerf/utils.py:1540-1600, for `--stage 1 --use_brdf
--use_restir`): stage-0 mesh + dataset views -> material field, environment map and vertex offsets.  Per iteration: one view, harness.render_stage1_outputs
(moved mesh -> BVH -> G-buffer -> ReSTIR frame under autograd -> tone curve -> dr.antialias -> SSAA), losses.stage1_loss with main.py's default weights,
losses.stage1_optimizer_step (three Adam optimisers: geometry lr_vert 1e-4, material 0.03, light 0.09; the x64 / /8 gradient rescaling; light clamp),
the reference's learning-rate schedules (main.py:285, nerf/utils.py:820-829), checkpoints in Trainer.save_checkpoint's layout (resume with --ckpt).

    python scripts/train_stage1.py --workspace <ws> --transforms <data>/transforms_train.json [--iters 7500 --spp 32 --ssaa 1 --downscale 1
        --bound 2 --roughness_min 0.08 --me_max 0 --scale 1 --offset 0 0 0 --ckpt <file.pth> --save_interval 500]
    python scripts/train_stage1.py --synthetic [--iters 60 --spp 8 --H 96 --W 96]      # fits a hidden synthetic scene; prints PSNR before / after
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train_stage1.py ...   # data parallel: one view per rank

Not here (the reference's Trainer / data loader / NeRF stage 0 are out of scope): the NeRF colour branch (`image`), mask / LPIPS / refine-error terms,
tensorboard, EMA, mesh export.  `np.random.seed` / `torch.manual_seed` (--seed) fix the view order, the jitter of the smoothness taps and the frame seeds."""
import argparse, json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, losses, raster, checkpoint as CK, dist as MD
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D


srgb_to_linear = harness.srgb_to_linear      # nerf/utils.py:57-58 (pinned to the reference's function in tests/test_losses.py)


def orbit_pose(az_deg, el_deg, dist=3.2):
    a, e = np.deg2rad(az_deg), np.deg2rad(el_deg)
    eye = dist * np.array([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)])
    fwd = -eye / np.linalg.norm(eye); right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right); up = np.cross(right, fwd)
    pose = np.eye(4, dtype=np.float32); pose[:3, :3] = np.stack([right, up, -fwd], 1); pose[:3, 3] = eye
    return pose


def synthetic_dataset(root, H, W, n_views=6, spp=64, mesh_error=0.0):
    """A hidden target scene (material field with seeded weights, sky with a sun) rendered through the HIP path into a NeRF-blender style folder."""
    from mirres_restir_nerf_mesh_amd import meters
    os.makedirs(os.path.join(root, "mesh_stage0"), exist_ok=True); os.makedirs(os.path.join(root, "train"), exist_ok=True)
    v, t = M.scene.make_mesh(4, 8)
    # the "stage-0" mesh the training starts from: the true one, or one with a smooth error (the object inflated and sheared by `mesh_error`) for
    # the vertex offsets to remove — the images below are always rendered from the true mesh
    v0 = v if mesh_error == 0 else (v * (1.0 + mesh_error * np.array([1.0, 0.6, 0.8], np.float32)) + mesh_error * 0.5 * v[:, [1, 2, 0]]).astype(np.float32)
    CK.write_ply(os.path.join(root, "mesh_stage0", "mesh_0.ply"), v0, t)
    aabb, mn, mx = CK.material_field_args(CK.material_config(bound=1.0))
    target = MLPTexture3D(aabb, channels=6, min_max=(mn.cuda(), mx.cuda()), seed=1)
    params, w0, w1, w2 = M.scene.make_matnet_params(seed=0)
    with torch.no_grad():
        target.encoder.params.copy_(torch.from_numpy(params).cuda())
        for i, w in zip((0, 2, 4), (w0, w1, w2)):
            target.net.net[i].weight.copy_(torch.from_numpy(w).cuda())
    env = torch.from_numpy(M.scene.make_env(64, 128)).cuda()
    Wk = RR.restirbvhWorker(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()); Wk.update_mesh(Wk.vrt, Wk.v_ind)
    focal = 0.5 * W / np.tan(0.5 * 0.6911); intr = (focal, focal, W * 0.5, H * 0.5)
    frames = []
    from PIL import Image
    for k in range(n_views):
        pose = orbit_pose(20.0 + 360.0 * k / n_views, 25.0 + 10.0 * (k % 2))
        img = harness.test_view(Wk, target, env, torch.from_numpy(pose), intr, H, W, spp, 1, random_offset=1000 + k)
        Image.fromarray((img.clamp(0, 1).cpu().numpy() * 255 + 0.5).astype(np.uint8)).save(os.path.join(root, "train", "r_%d.png" % k))
        frames.append({"file_path": "./train/r_%d" % k, "transform_matrix": pose.tolist()})
    tf = os.path.join(root, "transforms_train.json")
    json.dump({"camera_angle_x": 0.6911, "w": W, "h": H, "frames": frames}, open(tf, "w"))
    return tf


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--workspace"); p.add_argument("--transforms"); p.add_argument("--ckpt", default=None); p.add_argument("--synthetic", action="store_true")
    p.add_argument("--iters", type=int, default=7500); p.add_argument("--spp", type=int, default=32); p.add_argument("--ssaa", type=int, default=1); p.add_argument("--downscale", type=int, default=1)
    p.add_argument("--lr_vert", type=float, default=1e-4); p.add_argument("--learning_rate_mat", type=float, default=0.03); p.add_argument("--learning_rate_lgt", type=float, default=0.09)
    p.add_argument("--bound", type=float, default=None); p.add_argument("--roughness_min", type=float, default=None); p.add_argument("--me_max", type=float, default=None)
    p.add_argument("--cascade", type=int, default=None); p.add_argument("--light_probe_res_hw", type=int, nargs=2, default=[256, 512])
    p.add_argument("--scale", type=float, default=1.0); p.add_argument("--offset", type=float, nargs=3, default=[0.0, 0.0, 0.0])
    p.add_argument("--save_interval", type=int, default=500); p.add_argument("--seed", type=int, default=0); p.add_argument("--H", type=int, default=96); p.add_argument("--W", type=int, default=96)
    p.add_argument("--mesh_error", type=float, default=0.0, help="--synthetic: relative error of the starting mesh against the one the images show")
    p.add_argument("--freeze", nargs="*", default=[], choices=["geometry", "material", "light"], help="parameter groups left untouched (their learning rate set to 0)")
    p.add_argument("--pos_gradient_boost", type=float, default=1.0); p.add_argument("--lambda_extra_kd", type=float, default=0.0); p.add_argument("--quiet", action="store_true")
    a = p.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MIRRES_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}))
    np.random.seed(a.seed); torch.manual_seed(a.seed)
    if a.synthetic:
        a.workspace = a.workspace or os.path.join(ROOT, "gpurun_out", "train_ws")
        if rank == 0:
            synthetic_dataset(a.workspace, a.H, a.W, mesh_error=a.mesh_error)
        if world > 1:
            dist.barrier()
        a.transforms = os.path.join(a.workspace, "transforms_train.json"); a.bound = a.bound or 1.0
        a.light_probe_res_hw = [64, 128]
    if not (a.workspace and a.transforms):
        p.error("--workspace and --transforms are required (or --synthetic)")
    # ---- model: stage-0 mesh, material field, environment map (create_trainable_env_rnd(scale=0, bias=0.5), network.py:126), vertex offsets
    ck = CK.read_checkpoint(a.ckpt) if a.ckpt else None
    cfg = CK.resolve_material_config(ck.get("material_config") if ck else CK.material_config(bound=a.bound, roughness_min=a.roughness_min, me_max=a.me_max),
                                     bound=a.bound, roughness_min=a.roughness_min, me_max=a.me_max)
    v, t, _, _ = CK.load_stage0_mesh(a.workspace, a.cascade if a.cascade is not None else CK.cascade_of_bound(cfg["bound"]))
    aabb, mn, mx = CK.material_field_args(cfg)
    mlp = MLPTexture3D(aabb, channels=6, min_max=(mn.cuda(), mx.cuda()), seed=a.seed + 5)
    verts = torch.from_numpy(v).cuda(); tris = torch.from_numpy(t).cuda()
    voff = torch.zeros_like(verts).requires_grad_(True)
    env = torch.full((a.light_probe_res_hw[0], a.light_probe_res_hw[1], 3), 0.5, device="cuda").requires_grad_(True)
    step0 = 0
    if ck is not None:
        vo, lb = CK.apply_checkpoint(ck, mlp, n_vertices=v.shape[0])
        with torch.no_grad():
            if vo is not None: voff.copy_(vo)
            if lb is not None: env.copy_(lb)
        step0 = int(ck.get("global_step") or 0)
    Wk = RR.restirbvhWorker((verts + voff.detach()).contiguous(), tris); Wk.update_mesh(Wk.vrt, Wk.v_ind)
    topo = raster.antialias_topology(tris)
    # ---- data
    tf = json.load(open(a.transforms)); base = os.path.dirname(os.path.abspath(a.transforms))
    from PIL import Image
    frames = tf["frames"]
    first = np.asarray(Image.open(os.path.join(base, frames[0]["file_path"] + ".png")))
    H, Wd = first.shape[0] // a.downscale, first.shape[1] // a.downscale
    focal = 0.5 * Wd / np.tan(0.5 * tf["camera_angle_x"]); intr = (focal, focal, Wd * 0.5, H * 0.5)
    def load(fr):
        im = Image.open(os.path.join(base, fr["file_path"] + ".png"))
        if a.downscale > 1: im = im.resize((Wd, H), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(im).astype(np.float32) / 255.0).cuda().view(H * Wd, -1)
        lin = srgb_to_linear(x[:, :3])                                               # images_linear (utils.py:927)
        if x.shape[1] == 4:                                                          # white background (utils.py:948-955)
            return x[:, :3] * x[:, 3:] + (1 - x[:, 3:]), lin * x[:, 3:]
        return x[:, :3], lin
    data = [(torch.from_numpy(np.array(fr["transform_matrix"], np.float32)), load(fr)) for fr in frames]
    for pose, _ in data:
        pose[:3, 3] = pose[:3, 3] * a.scale + torch.tensor(a.offset)
    mods = RR.load_m_for_restir(Wd * a.ssaa, H * a.ssaa)
    # ---- optimisers and schedules (main.py:267-285, utils.py:820-829)
    o_geo = torch.optim.Adam([{"params": [voff], "lr": 0.0 if "geometry" in a.freeze else a.lr_vert, "weight_decay": 0}], eps=1e-15)
    o_mat = torch.optim.Adam([{"params": mlp.parameters(), "lr": 0.0 if "material" in a.freeze else a.learning_rate_mat}])
    o_lgt = torch.optim.Adam([{"params": [env], "lr": 0.0 if "light" in a.freeze else a.learning_rate_lgt}])
    s_geo = torch.optim.lr_scheduler.LambdaLR(o_geo, lambda it: 0.01 + 0.99 * (it / 500) if it <= 500 else 0.1 ** ((it - 500) / max(1, a.iters - 500)))
    brdf_sched = lambda it: max(0.0, 10 ** (-it * 0.0002))
    s_mat = torch.optim.lr_scheduler.LambdaLR(o_mat, brdf_sched); s_lgt = torch.optim.lr_scheduler.LambdaLR(o_lgt, brdf_sched)
    # resume (load_checkpoint, utils.py:1966-1990): optimiser moments and schedules continue where the checkpoint left them; a checkpoint without them
    # (the reference's default full=False files) still gets the SCHEDULES of its step — they are functions of the step count — and only the Adam moments restart
    scheds = {"lr_scheduler": s_geo, "scheduler_mat": s_mat, "scheduler_light": s_lgt}      # the reference's key names (nerf/utils.py:1863-1867)
    optims = {"optimizer": o_geo, "optimizer_mat": o_mat, "optimizer_light": o_lgt}
    if ck is not None:
        ts = ck.get("train_state") or {}
        for k_, o_ in optims.items():
            if k_ in ts:
                o_.load_state_dict(ts[k_])
        for k_, s_ in scheds.items():
            if k_ in ts:
                s_.load_state_dict(ts[k_])
            else:                                       # fast-forward: LambdaLR's rate is base_lr * lambda(step)
                s_.last_epoch = step0
                for g_, base, fn in zip(s_.optimizer.param_groups, s_.base_lrs, s_.lr_lambdas):
                    g_["lr"] = base * fn(step0)
                s_._last_lr = [g_["lr"] for g_ in s_.optimizer.param_groups]
        if rank == 0 and not all(k_ in ts for k_ in optims):
            print("[resume] %s holds no optimiser state: Adam moments restart at step %d (learning-rate schedules continue)" % (a.ckpt, step0), flush=True)
    opt = types.SimpleNamespace(use_brdf=True, lambda_extra_kd=a.lambda_extra_kd)
    sync = (lambda: MD.allreduce_gradients([voff] + list(mlp.parameters()) + [env])) if world > 1 else None

    def psnr_of(view):
        pose, (gt, _) = data[view]
        with torch.no_grad():
            img = harness.test_view(Wk, mlp, env.detach(), pose, intr, H, Wd, max(a.spp, 16), a.ssaa, random_offset=4242).view(-1, 3)
        return float(-10 * torch.log10(((img - gt) ** 2).mean()))
    psnr0 = psnr_of(0)
    order = np.random.permutation(len(data))
    t0 = time.perf_counter(); hist = []
    for it in range(step0, a.iters):
        k = (it * world + rank) % len(data)
        if k == 0 and rank == 0: order = np.random.permutation(len(data))
        pose, (gt, gt_lin) = data[int(order[k]) if world == 1 else k]
        for o in (o_geo, o_mat, o_lgt): o.zero_grad(set_to_none=True)
        out = harness.render_stage1_outputs(Wk, verts, voff, tris, mlp, env, mods, H, Wd, a.spp, a.ssaa, pose=pose, intrinsics=intr, topology=topo,
                                            pos_gradient_boost=a.pos_gradient_boost, with_normal_ao=a.lambda_extra_kd > 0)
        loss = losses.stage1_loss(out, gt, gt_lin, opt, vertices=verts, voffsets=voff, triangles=tris)
        val = losses.stage1_optimizer_step(loss, o_geo, o_mat, o_lgt, light_base=env, encoder_params=mlp.encoder.params, scheduler=s_geo, scheduler_mat=s_mat,
                                           scheduler_light=s_lgt, grad_sync=sync)
        hist.append(val)
        if rank == 0 and not a.quiet and (it % 20 == 0 or it == a.iters - 1):
            print("[%5d/%d] loss %.5f  lr vert %.2e mat %.2e light %.2e" % (it, a.iters, val, o_geo.param_groups[0]["lr"], o_mat.param_groups[0]["lr"], o_lgt.param_groups[0]["lr"]), flush=True)
        if rank == 0 and a.save_interval > 0 and ((it + 1) % a.save_interval == 0 or it == a.iters - 1):
            os.makedirs(os.path.join(a.workspace, "checkpoints"), exist_ok=True)
            CK.save_checkpoint(os.path.join(a.workspace, "checkpoints", "ngp_stage1_ep%04d.pth" % (it + 1)), mlp, voff, env, epoch=it + 1, global_step=it + 1, material_config=cfg,
                               train_state={**{k_: o_.state_dict() for k_, o_ in optims.items()}, **{k_: s_.state_dict() for k_, s_ in scheds.items()}})
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    Wk.update_mesh((verts + voff.detach()).contiguous(), tris)
    psnr1 = psnr_of(0)
    if rank == 0:
        n = max(1, a.iters - step0)
        print("stage-1 training %d iterations at %dx%d ssaa %d spp %d on %d GPU(s): %.1f ms/iteration; loss %.5f -> %.5f; PSNR of view 0: %.2f -> %.2f dB" % (
            n, Wd, H, a.ssaa, a.spp, world, 1e3 * dt / n, float(np.mean(hist[:5])) if hist else float("nan"), float(np.mean(hist[-5:])) if hist else float("nan"), psnr0, psnr1))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
