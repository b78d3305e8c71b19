"""`--test --spp N` evaluation of a reference workspace on the HIP path: stage-0 mesh (+ stage-1 vertex offsets), material field and
environment map from a reference checkpoint, dataset cameras from a NeRF-blender `transforms_*.json`, one fused frame per view, PNG results and
PSNR / SSIM against the dataset images when they can be read.

    python scripts/evaluate.py --workspace <ws> --ckpt <ws>/checkpoints/ngp_stage1_ep0100.pth --transforms <data>/transforms_test.json \
        [--spp 512 --ssaa 2 --downscale 1 --bound 2 --roughness_min 0.08 --me_max 0 --out <ws>/results_brdf --limit 0 --synthetic]

The material-field constants are the reference's CLI ones and MUST equal the training run's (nerf/network.py:119-125): `--bound` (hash-grid AABB
= +-bound, main.py:39 default 2; also the mesh cascade count 1 + ceil(log2(bound)), nerf/renderer.py:97, unless `--cascade` overrides it),
`--roughness_min` / `--me_max` (main.py:109-110,169-170) and, for completeness, `--kd_min` / `--kd_max` (main.py:167-168).  Checkpoints written by
this package record them (`material_config`); a flag given on the command line wins, a mismatch with the recorded value is reported.

Relighting (reference: `--test --envmap_path X.hdr --albedo_scale_x/y/z`, nerf/network.py:134-139, nerf/renderer.py:1025-1026, 1086-1089, 1109-1111):
`--envmap_path` replaces the trained environment map by a Radiance .hdr file (any size) and switches the albedo scaling on.

Several GPUs (`python -m torch.distributed.run --nproc-per-node N scripts/evaluate.py ... --shard views|strips|spp`): `views` (default) deals the
dataset views round-robin to the ranks — no data-path communication, metrics reduced at the end; `strips` / `spp` render EVERY view on all ranks
(exact row strips with per-sample halo exchange and an all-gather of the radiance rows / sample slices with one all-reduce; dist.py) — what a single
large frame wants.  Rank 0 writes the images of the frames it holds (all of them for strips / spp).

`--synthetic` builds a throw-away workspace (synthetic mesh, random material field, sky map, four orbit cameras) first and evaluates that — the
smoke run of this script on a box without a reference workspace.  Camera convention: the blender `transform_matrix` is the cam2world pose with its
translation scaled by `--scale` and shifted by `--offset` (nerf_matrix_to_ngp, nerf/provider.py:18-21; pass the values the reference run used)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mirres_restir_nerf_mesh_amd as M
from mirres_restir_nerf_mesh_amd import renderer_restir as RR, harness, checkpoint as CK, meters, dist as MD
from mirres_restir_nerf_mesh_amd.render_helper import MLPTexture3D


def nerf_pose(pose, scale=1.0, offset=(0.0, 0.0, 0.0)):
    """nerf_matrix_to_ngp of this reference (nerf/provider.py:18-21): the rotation is kept, the translation is scaled and shifted."""
    p = np.array(pose, np.float32)
    p[:3, 3] = p[:3, 3] * scale + np.asarray(offset, np.float32)
    return p


def synthetic_workspace(root, H=100, W=100):
    os.makedirs(os.path.join(root, "mesh_stage0"), exist_ok=True); os.makedirs(os.path.join(root, "checkpoints"), exist_ok=True)
    v, t = M.scene.make_mesh(5, 16)
    CK.write_ply(os.path.join(root, "mesh_stage0", "mesh_0.ply"), v, t)
    mn, mx = M.scene.material_min_max()
    torch.manual_seed(0)
    mlp = MLPTexture3D(torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32), channels=6, min_max=(torch.from_numpy(mn).cuda(), torch.from_numpy(mx).cuda()), seed=1)
    with torch.no_grad():
        mlp.encoder.params.mul_(2e3)
    ck = os.path.join(root, "checkpoints", "ngp_stage1_ep0001.pth")
    CK.save_checkpoint(ck, mlp, torch.zeros(v.shape[0], 3), torch.from_numpy(M.scene.make_env(64, 128)), epoch=1,
                       material_config=CK.material_config(bound=1.0))
    frames = []
    for k in range(4):
        az, el = np.deg2rad(30.0 + 90.0 * k), np.deg2rad(30.0)
        eye = 3.2 * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        fwd = -eye / np.linalg.norm(eye); right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right); up = np.cross(right, fwd)
        pose = np.eye(4); pose[:3, :3] = np.stack([right, up, -fwd], 1); pose[:3, 3] = eye
        frames.append({"file_path": "./test/r_%d" % k, "transform_matrix": pose.tolist()})
    tf = os.path.join(root, "transforms_test.json")
    json.dump({"camera_angle_x": 0.6911, "w": W, "h": H, "frames": frames}, open(tf, "w"))
    return ck, tf


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--workspace"); p.add_argument("--ckpt"); p.add_argument("--transforms"); p.add_argument("--out")
    p.add_argument("--spp", type=int, default=512); p.add_argument("--ssaa", type=int, default=2); p.add_argument("--downscale", type=int, default=1)
    p.add_argument("--cascade", type=int, default=None, help="mesh cascades; default 1 + ceil(log2(bound)) as nerf/renderer.py:97")
    p.add_argument("--bound", type=float, default=None, help="main.py --bound (default 2): material-field AABB = +-bound")
    p.add_argument("--roughness_min", type=float, default=None, help="main.py --roughness_min (default 0.08)")
    p.add_argument("--me_max", type=float, default=None, help="main.py --me_max (default 0.0)")
    p.add_argument("--kd_min", type=float, nargs=3, default=None); p.add_argument("--kd_max", type=float, nargs=3, default=None)
    p.add_argument("--limit", type=int, default=0); p.add_argument("--H", type=int, default=800); p.add_argument("--W", type=int, default=800)
    p.add_argument("--save_maps", action="store_true", help="also write kd / ks / normal / env_map / diffuse and specular light as EXR files, like Trainer.test (nerf/utils.py:1372-1377)")
    p.add_argument("--scale", type=float, default=1.0); p.add_argument("--offset", type=float, nargs=3, default=[0.0, 0.0, 0.0]); p.add_argument("--synthetic", action="store_true")
    p.add_argument("--envmap_path", default="None", help="main.py --envmap_path: Radiance .hdr environment map for relighting")
    p.add_argument("--albedo_scale_x", type=float, default=1.0); p.add_argument("--albedo_scale_y", type=float, default=1.0); p.add_argument("--albedo_scale_z", type=float, default=1.0)
    p.add_argument("--shard", choices=("views", "strips", "spp"), default="views", help="how N > 1 ranks (torch.distributed.run) divide the work")
    p.add_argument("--lpips_vgg", default=None, help="torchvision vgg16 state dict (or a state dict of lpips.LPIPS): adds the reference's LPIPS (vgg) meter; no weights ship with the image")
    p.add_argument("--lpips_lin", default=None, help="the lpips package's weights/v0.1/vgg.pth (the five linear heads)")
    a = p.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MIRRES_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", torch.cuda.current_device())} if backend == "nccl" else {}))
    if a.synthetic:
        a.workspace = a.workspace or os.path.join(ROOT, "gpurun_out", "eval_ws")
        if rank == 0:
            a.ckpt, a.transforms = synthetic_workspace(a.workspace, a.H // a.downscale, a.W // a.downscale)
        if world > 1:
            dist.barrier()
        a.ckpt, a.transforms = os.path.join(a.workspace, "checkpoints", "ngp_stage1_ep0001.pth"), os.path.join(a.workspace, "transforms_test.json")
    if not (a.workspace and a.ckpt and a.transforms):
        p.error("--workspace, --ckpt and --transforms are required (or --synthetic)")
    out_dir = a.out or os.path.join(a.workspace, "results_brdf")
    ck = CK.read_checkpoint(a.ckpt)
    cfg = CK.resolve_material_config(ck.get("material_config"), bound=a.bound, roughness_min=a.roughness_min, me_max=a.me_max, kd_min=a.kd_min, kd_max=a.kd_max)
    cascade = a.cascade if a.cascade is not None else CK.cascade_of_bound(cfg["bound"])
    v, t, v_cumsum, _ = CK.load_stage0_mesh(a.workspace, cascade)
    aabb, mn, mx = CK.material_field_args(cfg)     # nerf/network.py:119-125
    mlp = MLPTexture3D(aabb, channels=6, min_max=(mn.cuda(), mx.cuda()))
    voff, light = CK.apply_checkpoint(ck, mlp, n_vertices=v.shape[0])
    albedo_scale = None
    if a.envmap_path != "None":      # relighting: the external map as it is (no clamp), albedo scaling on (network.py:134-139, renderer.py:1109-1111)
        light = torch.from_numpy(np.ascontiguousarray(harness.read_hdr(a.envmap_path))).cuda()
        albedo_scale = (a.albedo_scale_x, a.albedo_scale_y, a.albedo_scale_z)
    if light is None:
        raise SystemExit("%s has no light_base (a --use_brdf stage-1 checkpoint is needed)" % a.ckpt)
    verts = torch.from_numpy(v).cuda() + (voff if voff is not None else 0)
    W = RR.restirbvhWorker(verts.contiguous(), torch.from_numpy(t).cuda()); W.update_mesh(W.vrt, W.v_ind)
    tf = json.load(open(a.transforms)); base = os.path.dirname(os.path.abspath(a.transforms))
    Hh, Ww = int(tf.get("h", a.H)) // a.downscale, int(tf.get("w", a.W)) // a.downscale
    focal = 0.5 * Ww / np.tan(0.5 * tf["camera_angle_x"])
    intr = (focal, focal, Ww * 0.5, Hh * 0.5)
    frames = tf["frames"][: a.limit] if a.limit > 0 else tf["frames"]
    pm, sm = meters.PSNRMeter(), meters.SSIMMeter()
    lm = meters.LPIPSMeter(vgg=a.lpips_vgg, lin=a.lpips_lin) if a.lpips_vgg else None        # main.py:250: [PSNRMeter(), SSIMMeter(), LPIPSMeter(device=device)]
    name = os.path.splitext(os.path.basename(a.ckpt))[0]
    if world == 1 or a.shard == "views":
        # the engine context of the frame size and its batch pool (tens of GB: one hipMalloc + clear) are set up here, not inside the first view's timed region
        from mirres_restir_nerf_mesh_amd._ops import get_ctx
        get_ctx(Ww * a.ssaa, Hh * a.ssaa).reserve()
    t_render = 0.0
    n_mine = 0
    # exact strip sharding: the boundaries follow the strips' measured times from view to view (a test set's cameras move smoothly; same pixels for any boundaries)
    balancer = MD.StripBalancer(Hh * a.ssaa, world) if (world > 1 and a.shard == "strips") else None
    for i, fr in enumerate(frames):
        if world > 1 and a.shard == "views" and i % world != rank:
            continue
        n_mine += 1
        pose = nerf_pose(fr["transform_matrix"], a.scale, a.offset)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img = harness.test_view(W, mlp, light, torch.from_numpy(pose), intr, Hh, Ww, a.spp, a.ssaa, random_offset=i * 7919, albedo_scale=albedo_scale,
                                shard=a.shard if world > 1 and a.shard != "views" else None, rank=rank, world=world, return_maps=a.save_maps, balancer=balancer)
        maps = None
        if a.save_maps:
            img, maps = img
        torch.cuda.synchronize(); t_render += time.perf_counter() - t0
        if world > 1 and a.shard != "views" and rank != 0:
            continue                                                       # every rank holds the whole frame; rank 0 writes and scores it
        files = meters.write_test_frame(out_dir, name, i, img)
        if maps is not None:                                               # kd / ks / normal / env_map / diffuse + specular light as EXR (utils.py:1372-1377)
            files += meters.write_test_maps(out_dir, name, i, maps)
        gt_path = os.path.join(base, fr["file_path"] + ".png")
        note = ""
        if os.path.exists(gt_path):
            from PIL import Image
            gt = np.asarray(Image.open(gt_path).resize((Ww, Hh), Image.BILINEAR)).astype(np.float32) / 255.0
            if gt.shape[-1] == 4:
                gt = gt[..., :3] * gt[..., 3:] + (1 - gt[..., 3:])                      # white background, as the reference's loader composes it
            gt_t = torch.from_numpy(gt).cuda()
            note = "  PSNR %.3f  SSIM %.4f" % (pm.update(img, gt_t), sm.update(img, gt_t))
            if lm is not None:
                note += "  LPIPS %.4f" % lm.update(img, gt_t)
        print("[%d/%d] %s%s" % (i + 1, len(frames), os.path.basename(files[0]), note), flush=True)
    n = len(frames)
    if world > 1:      # wall time of the job = the slowest rank; metric sums over the ranks that scored frames
        red = torch.tensor([t_render, pm.V, float(pm.N), sm.V, float(sm.N), lm.V if lm else 0.0, float(lm.N) if lm else 0.0], dtype=torch.float64, device="cuda")
        tmax = red[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(red, op=dist.ReduceOp.SUM)
        t_render = float(tmax.item()); pm.V, pm.N, sm.V, sm.N = float(red[1]), int(red[2]), float(red[3]), int(red[4])
        if lm:
            lm.V, lm.N = float(red[5]), int(red[6])
    if rank == 0:
        print("rendered %d views %dx%d ssaa %d spp %d on %d GPU(s)%s: %.1f ms/view (%.1f Msamples/s)" % (
            n, Ww, Hh, a.ssaa, a.spp, world, (" [%s]" % a.shard) if world > 1 else "", 1e3 * t_render / max(n, 1), n * Ww * Hh * a.ssaa ** 2 * a.spp / max(t_render, 1e-9) / 1e6))
        if pm.N:
            print(pm.report(), sm.report(), lm.report() if lm and lm.N else "")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
