"""Harness pieces around the hot path (SURVEY §8 a-H): synthetic G-buffer by primary-ray cast through the engine's BVH,
and the reference's post-processing formulas (sRGB, SSAA down-scale, PSNR) restated for the evaluation driver.
  linear2srgb : nerf/utils.py:80-106   x<=0.0031308 ? 12.92x : 1.055 (x+1e-6)^(1/2.4) - 0.055
  PSNR        : nerf/utils.py:611-623  -10 log10(mean((pred-gt)^2))
"""
import numpy as np
import torch

from . import scene


def build_gbuffer(worker, H, W, ssaa=1, azimuth_deg=30.0, elevation_deg=30.0, kd=(0.6, 0.6, 0.6), roughness=0.5, metallic=0.0, mlp_mat=None):
    """Primary visibility from the BVH closest-hit kernel (stands in for nvdiffrast). Returns a dict of [N,*] CUDA tensors at the
    internal (ssaa-scaled) resolution, in the layout render_stage1 hands to run_restir_di_with_pt (nerf/renderer.py:1083-1123)."""
    h, w = H * ssaa, W * ssaa
    eye, rd = scene.camera_rays(h, w, azimuth_deg, elevation_deg)
    dev = worker.vrt.device
    rays_d = torch.from_numpy(rd).to(dev)
    rays_o = torch.from_numpy(eye).to(dev)[None].expand(h * w, 3).contiguous()
    r = worker.trace(rays_o, rays_d, closest=True)
    occ = r["hit"].to(torch.float32)[:, None].contiguous()
    pos = r["pos"]
    normal = torch.where(occ > 0.5, r["normal"], torch.zeros_like(r["normal"])).contiguous()
    depth = torch.norm(pos - rays_o, dim=1, keepdim=True).contiguous()
    N = h * w
    if mlp_mat is not None:
        kdks = mlp_mat.sample_no_di(pos)
        kd_map = kdks[:, 0:3].contiguous()
        rm = torch.cat((kdks[:, 4:5], kdks[:, 5:6]), dim=-1).contiguous()
    else:
        kd_map = torch.tensor(kd, dtype=torch.float32, device=dev)[None].expand(N, 3).contiguous()
        rm = torch.tensor([roughness, metallic], dtype=torch.float32, device=dev)[None].expand(N, 2).contiguous()
    return dict(fx=w, fy=h, occ=occ, pos=pos.contiguous(), normal=normal, depth=depth, kd=kd_map, rm=rm, ray_dir=rays_d, eye=eye)


def build_gbuffer_stage1(worker, vertices, triangles, H, W, ssaa=1, azimuth_deg=30.0, elevation_deg=30.0, mlp_mat=None, pose=None, intrinsics=None):
    """The front half of render_stage1 (nerf/renderer.py:978-1022, 1083-1094) on the engine's own operators: raster record by dr.rasterize's own call shape
    (raster.dr.rasterize(glctx, vertices_clip, triangles, (h, w)) for a dataset camera; raster.rasterize_raycast for the synthetic orbit view), xyzs / smooth normal / geometric normal by raster.interpolate (dr.interpolate), auto_normals,
    renderutils.prepare_shading_normal, safe_normalize, material field. Differentiable w.r.t. `vertices` through the interpolations and the material
    field's position gradient. Returns the dict build_gbuffer returns, plus `rast`.  Camera: the synthetic orbit view (azimuth / elevation), or a
    dataset camera (`pose` [4,4] cam2world + `intrinsics` (fx, fy, cx, cy) at H x W) — then the dict also carries `mvp` (mvp_from_pose at the internal
    resolution) and `vertices_clip` [V,4] (:981), what dr.antialias needs."""
    from . import raster
    from .renderutils.ops import prepare_shading_normal
    h, w = H * ssaa, W * ssaa
    dev = vertices.device
    cam = {}
    if pose is not None:
        pose = pose.to(dev)
        fx_, fy_, cx_, cy_ = (float(v) * ssaa for v in intrinsics)
        rays_o, rays_d = get_rays(pose, (fx_, fy_, cx_, cy_), h, w)
        eye_t = pose[:3, 3].to(torch.float32).contiguous(); eye = eye_t.detach().cpu().numpy()
        mvp = mvp_from_pose(pose, (fx_, fy_, cx_, cy_), h, w)
        cam = dict(mvp=mvp, vertices_clip=torch.cat((vertices, torch.ones_like(vertices[:, :1])), dim=1) @ mvp.t())      # :981
        # rast, rast_out_deriv_s = dr.rasterize(self.glctx, vertices_clip, self.triangles, (h, w))                          # :983 — the reference's call, unchanged
        # (the exact camera is in hand: `mvp=` spares rasterize the float64 re-fit of the matrix — a host synchronisation per view)
        rast4, rast_db = raster.dr.rasterize(raster.RasterizeContext(worker), cam["vertices_clip"].unsqueeze(0), triangles, (h, w), mvp=mvp)
        rast = rast4.view(h * w, 4)
        cam["rast_db"] = rast_db
    else:
        eye, rd = scene.camera_rays(h, w, azimuth_deg, elevation_deg)
        rays_d = torch.from_numpy(rd).to(dev)
        eye_t = torch.from_numpy(eye).to(dev)
        rays_o = eye_t[None].expand(h * w, 3).contiguous()
        rast = raster.rasterize_raycast(worker, rays_o, rays_d)      # the synthetic orbit view has no projection matrix: the same record from explicit rays
    tri = triangles.to(torch.int32)
    xyzs = raster.interpolate(vertices, rast, tri)                                                         # :985
    v_nrm, t_nrm_idx = raster.auto_normals(vertices, tri)                                                  # :978
    v0, v1, v2 = (vertices[tri[:, k].long(), :] for k in range(3))
    fn = torch.cross(v1 - v0, v2 - v0, dim=-1)
    face_normals = fn / torch.sqrt(torch.clamp(torch.sum(fn * fn, -1, keepdim=True), min=1e-20))           # safe_normalize (:994)
    face_idx = torch.arange(0, tri.shape[0], dtype=torch.int32, device=dev)[:, None].repeat(1, 3)
    gb_geometric_normal = raster.interpolate(face_normals, rast, face_idx)                                 # :996
    gb_normal = raster.interpolate(v_nrm, rast, t_nrm_idx)                                                 # :998
    gb_tangent = torch.zeros_like(gb_normal)
    sh = lambda x: x.view(1, h, w, 3)
    nrm = prepare_shading_normal(sh(xyzs), eye_t.view(1, 1, 1, 3), None, sh(gb_normal), sh(gb_tangent), sh(gb_geometric_normal), two_sided_shading=True, opengl=True)  # :1013
    nrm = nrm.reshape(-1, 3)
    nrm = nrm / torch.sqrt(torch.clamp(torch.sum(nrm * nrm, -1, keepdim=True), min=1e-20))                 # :1014
    occ = (rast[:, 3:4] > 0).float()                                                                       # mask (:1028)
    N = h * w
    kdks = mlp_mat.sample(xyzs) if mlp_mat is not None else None
    if kdks is not None:
        kd_map = kdks[:, 0:3].contiguous(); rm = torch.cat((kdks[:, 4:5], kdks[:, 5:6]), dim=-1).contiguous()   # :1020, :1092
    else:
        kd_map = torch.full((N, 3), 0.6, device=dev); rm = torch.tensor([0.5, 0.0], device=dev)[None].expand(N, 2).contiguous()
    depth = torch.norm(xyzs - rays_o, dim=1, keepdim=True)                                                 # :1096
    return dict(fx=w, fy=h, occ=occ, pos=xyzs, normal=nrm * occ, depth=depth, kd=kd_map, rm=rm, ray_dir=rays_d, eye=eye, rast=rast, **cam)


def get_rays(pose, intrinsics, H, W):
    """Full-frame camera rays of nerf/utils.py:350-423 (N = -1): pixel centres (i + 0.5, j + 0.5), camera looks down -z with y flipped, directions
    NOT normalised (their z component is -1 in camera space), origin = the pose's translation.  `pose` [4, 4] cam2world, `intrinsics` (fx, fy, cx, cy).
    Returns (rays_o [H*W, 3], rays_d [H*W, 3]) in row-major pixel order."""
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    dev = pose.device
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev) + 0.5, torch.arange(W, dtype=torch.float32, device=dev) + 0.5, indexing="ij")
    cam = torch.stack(((i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)), dim=-1).view(-1, 3)
    rays_d = cam @ pose[:3, :3].to(torch.float32).t()
    rays_o = pose[:3, 3].to(torch.float32)[None].expand_as(rays_d)
    return rays_o.contiguous(), rays_d.contiguous()


def mvp_from_pose(pose, intrinsics, H, W, near=0.05, far=1000.0):
    """The model-view-projection matrix the dataset hands to render_stage1 (nerf/provider.py:277-288): an OpenGL-style perspective projection with
    the image's y axis flipped (row 0 at the top, like get_rays) times the inverse of the cam2world pose.  `near` = --min_near (main.py default 0.05),
    far = 1000.  vertices_clip = pad(vertices, 1) @ mvp.T (nerf/renderer.py:981); pixel (i, j)'s centre is NDC ((2i + 1) / W - 1, (2j + 1) / H - 1)."""
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    y = H / (2.0 * fy)
    aspect = W / H
    proj = torch.tensor([[1 / (y * aspect), 0, 0, 0], [0, -1 / y, 0, 0], [0, 0, -(far + near) / (far - near), -(2 * far * near) / (far - near)], [0, 0, -1, 0]],
                        dtype=torch.float32, device=pose.device)
    return proj @ torch.inverse(pose.to(torch.float32))


def view_dirs(rays_d, H, W, ssaa=1):
    """The per-pixel view directions render_stage1 shades with (nerf/renderer.py:935-946): with --ssaa > 1 the H x W directions are magnified
    with nearest-neighbour lookup (every ssaa x ssaa block shares its output pixel's direction), then safe_normalize."""
    d = rays_d.view(H, W, 3)
    if ssaa > 1:
        d = d.repeat_interleave(ssaa, dim=0).repeat_interleave(ssaa, dim=1)
    d = d.reshape(-1, 3)
    return (d / torch.sqrt(torch.clamp(torch.sum(d * d, -1, keepdim=True), min=1e-20))).contiguous()


def build_gbuffer_from_pose(worker, pose, intrinsics, H, W, ssaa=1, mlp_mat=None, kd=(0.6, 0.6, 0.6), roughness=0.5, metallic=0.0):
    """build_gbuffer for a dataset camera: primary visibility by casting the pixel-centre rays of the internal (ssaa-scaled) frame through the
    BVH (intrinsics scaled by ssaa, which is where nvdiffrast's raster samples sit), shading directions as the reference forms them (view_dirs)."""
    h, w = H * ssaa, W * ssaa
    dev = worker.vrt.device
    pose = pose.to(dev)
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    rays_o, rays_hi = get_rays(pose, (fx * ssaa, fy * ssaa, cx * ssaa, cy * ssaa), h, w)
    rays_hi = rays_hi / torch.norm(rays_hi, dim=1, keepdim=True)
    r = worker.trace(rays_o, rays_hi.contiguous(), closest=True)
    occ = r["hit"].to(torch.float32)[:, None].contiguous()
    pos = r["pos"].contiguous()
    normal = torch.where(occ > 0.5, r["normal"], torch.zeros_like(r["normal"])).contiguous()
    depth = torch.norm(pos - rays_o, dim=1, keepdim=True).contiguous()
    N = h * w
    if mlp_mat is not None:
        kdks = mlp_mat.sample_no_di(pos)
        kd_map = kdks[:, 0:3].contiguous(); rm = torch.cat((kdks[:, 4:5], kdks[:, 5:6]), dim=-1).contiguous()
    else:
        kd_map = torch.tensor(kd, dtype=torch.float32, device=dev)[None].expand(N, 3).contiguous()
        rm = torch.tensor([roughness, metallic], dtype=torch.float32, device=dev)[None].expand(N, 2).contiguous()
    dirs = view_dirs(get_rays(pose, (fx, fy, cx, cy), H, W)[1], H, W, ssaa)
    return dict(fx=w, fy=h, occ=occ, pos=pos, normal=normal, depth=depth, kd=kd_map, rm=rm, ray_dir=dirs, eye=pose[:3, 3].detach().cpu().numpy())


def test_view(worker, mlp_mat, env_map, pose, intrinsics, H, W, spp, ssaa=1, random_offset=0, de=2, c=2.0, n=0.1, p=0.001, max_bounce=None,
              albedo_scale=None, shard=None, rank=0, world=1, group=None, return_maps=False, balancer=None):
    """One `--test --spp N` frame of the BRDF branch (Trainer.test_step -> render_stage1(is_test=True), nerf/renderer.py:1083-1129, 1162-1164,
    1208-1209, 1265-1302): G-buffer for the dataset camera, the fused frame (mirres_render), tone curve, alpha, SSAA down-scale, white background.
    Returns the [H, W, 3] image in [0, 1]; with `return_maps` also the dict of float maps that Trainer.test saves as EXR files (meters.write_test_maps).
    Relighting (`--envmap_path`, :1025-1026, 1086-1089, 1109-1111): pass the external map as `env_map` and `albedo_scale` = (--albedo_scale_x, _y,
    _z): the primary albedo is scaled here and `use_scale` does the same at the indirect hits.
    One view on several GPUs: `shard` = "strips" (exact row strips + halo exchange + all-gather of radiance rows, dist.render_strips) or "spp"
    (sample slices + all-reduce, dist.render_sharded) with this process's `rank` of `world`; every rank returns the whole image. `balancer`: a
    dist.StripBalancer kept by the caller across the views of a run — strip boundaries then follow the strips' measured times (same pixels for any boundaries)."""
    from . import renderer_restir as RR
    from . import dist as MD
    from ._ops import get_ctx
    g = build_gbuffer_from_pose(worker, pose, intrinsics, H, W, ssaa, mlp_mat)
    use_scale = albedo_scale is not None
    scale = tuple(float(x) for x in albedo_scale) if use_scale else (1.0, 1.0, 1.0)
    if use_scale:
        g["kd"] = (g["kd"] * torch.tensor(scale, dtype=torch.float32, device=g["kd"].device)[None, :]).contiguous()      # :1086-1089
    ctx = get_ctx(g["fx"], g["fy"]) if max_bounce is None else get_ctx(g["fx"], g["fy"], max_bounce=max_bounce)
    if world > 1 and shard == "strips":
        out = MD.render_strips(ctx, worker, mlp_mat, env_map, g, spp, random_offset, rank, world, de, 2 ** (de - 1), c, n, p, use_scale, scale, group, max_bounce, balancer=balancer)
    elif world > 1 and shard == "spp":
        out = MD.render_sharded(ctx, worker, mlp_mat, env_map, g, spp, random_offset, rank, world, de, 2 ** (de - 1), c, n, p, use_scale, scale, group)
    else:
        out = RR.render_fused(ctx, worker, mlp_mat, use_scale, scale, env_map, g["occ"].clone(), g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"],
                              spp, de, 2 ** (de - 1), c, n, p, random_offset)[0]
    img = postprocess(torch.nan_to_num(out[0], 0.0), g["occ"], H, W, ssaa)
    if not return_maps:
        return img
    # the float maps Trainer.test writes next to the image (preds_brdf_list of render_stage1, nerf/renderer.py:1310-1330; utils.py:1372-1377): albedo, (0, roughness,
    # metallic), shading normal, environment map, denoised diffuse / specular light — at the internal (ssaa) resolution, background zero
    h, w = g["fy"], g["fx"]
    on = g["occ"]
    m3 = lambda x: (x * on).view(h, w, 3)
    ks = torch.cat((torch.zeros_like(g["rm"][:, :1]), g["rm"]), dim=1)
    maps = dict(kd=m3(g["kd"]), ks=m3(ks), normal=m3(g["normal"]), env_map=env_map.detach(), rgb_diffuse_light=m3(out[1]), rgb_specular_light=m3(out[2]))
    return img, maps


test_view.__test__ = False      # not a pytest case


def render_stage1_outputs(worker, vertices, voffsets, triangles, mlp_mat, env_map, mods, H, W, spp, ssaa=1, azimuth_deg=30.0, elevation_deg=30.0,
                          jitter_std=0.01, bg_color=1.0, gb_depth=None, with_normal_ao=False, de=2, c=2.0, n=0.1, p=0.001, pose=None, intrinsics=None,
                          topology=None, pos_gradient_boost=1.0):
    """`render_stage1` for `--stage 1 --use_brdf --use_restir` training (nerf/renderer.py:960-1302) as far as the material / light / geometry
    branch goes: moved mesh -> BVH update -> G-buffer front half (build_gbuffer_stage1) -> jittered material taps (:1016-1022) ->
    run_restir_di_with_pt (:1112-1123) -> clamp, tone curve (:1125-1129, 1162-1164) -> dr.antialias of alpha and of every output image (:1184-1200;
    the indirect images detached, as there) -> alpha (:1208-1240) -> SSAA down-scale to H x W when ssaa > 1 (:1265-1297) -> background (:1301).  Returns the entries of the reference's `outputs` dict that losses.stage1_loss
    reads, except `image` (the NeRF colour branch belongs to stage 0).  With a dataset camera (`pose`, `intrinsics`) the antialias step runs and the
    image loss reaches the vertex positions through visibility; with the synthetic orbit camera (no projection matrix) it is skipped.  `topology` =
    raster.antialias_topology(triangles) (built per call when None).  `gb_depth` ([N, 2]: z, |dz|; :1070-1081) selects the --use_bi_de bilateral
    finish, None the a-trous one.  (normal_grad, which needs the perturbed-normal texture the --use_brdf path never enables, is zero)."""
    from . import renderer_restir as RR
    from . import raster
    moved = vertices + voffsets
    worker.update_mesh(moved.detach().contiguous(), triangles)
    g = build_gbuffer_stage1(worker, moved, triangles, H, W, ssaa, azimuth_deg, elevation_deg, mlp_mat, pose=pose, intrinsics=intrinsics)
    fx, fy = g["fx"], g["fy"]; N = fx * fy
    dev = moved.device
    xyzs = g["pos"]
    tex_jitter = mlp_mat.sample(xyzs + torch.normal(mean=0.0, std=jitter_std, size=xyzs.shape, device=dev))
    tex = mlp_mat.sample(xyzs)
    kd_grad = torch.abs(tex_jitter[:, 0:3] - tex[:, 0:3])
    ks_grad = torch.abs(tex_jitter[:, 3:6] - tex[:, 3:6]) * torch.tensor([0.0, 1.0, 1.0], device=dev)[None, :]       # the o-component is left out
    z = lambda *s: torch.zeros(s, device=dev)
    out = RR.run_restir_di_with_pt(False, 1.0, 1.0, 1.0, mlp_mat, gb_depth, worker, *mods[:8], *mods[8:17], env_map, g["occ"].clone(), g["normal"],
                                   g["depth"], g["kd"], g["rm"], g["ray_dir"], xyzs.detach(), z(N, 1), z(N, 4), z(N, 3), z(N, 3), fx, fy, spp, de,
                                   2 ** (de - 1), c, n, p)
    extra = {}
    out_ao = None
    if with_normal_ao:      # --lambda_extra_kd > 0 (:1150-1158, 1226-1228): the kernel is launched off the denoising module handle, as the reference does
        out_ao = torch.zeros((N, 3), dtype=torch.float32, device=dev)
        mods[7].process_normal_ao(framedim_x=int(fx), framedim_y=int(fy), occ_map=g["occ"], normal_map=g["normal"].detach().contiguous(), ray_dir=g["ray_dir"], out_ao=out_ao) \
            .launchRaw(blockSize=(16, 16, 1), gridSize=((int(fx) + 15) // 16, (int(fy) + 15) // 16, 1))
    alpha = g["occ"]
    brdf_rgbs = linear2srgb(torch.clamp(torch.nan_to_num(out[0], 0.0), 0.0, 1.0))                                   # :1125-1129, 1162-1164
    if "vertices_clip" in g:
        tri32 = triangles.to(torch.int32)
        topo = topology if topology is not None else raster.antialias_topology(tri32)
        rast4 = g["rast"].view(1, fy, fx, 4); clip = g["vertices_clip"][None]
        aa = lambda x: raster.antialias(x.view(1, fy, fx, x.shape[-1]), rast4, clip, tri32, topology_hash=topo, pos_gradient_boost=pos_gradient_boost) \
            .view(N, x.shape[-1]).clamp(0, 1)                                                                        # :1184-1200
        alpha = aa(alpha)
    else:
        aa = lambda x: torch.clamp(x, 0.0, 1.0)
    if out_ao is not None:
        extra["normal_ao"] = (alpha * aa(out_ao.detach())).detach()                                                 # :1195, 1226-1228
    res = dict(image_brdf=alpha * aa(brdf_rgbs), diffuse_light=alpha * aa(out[1]), specular_light=alpha * aa(out[2]),                      # :1186, 1209-1240
               img_brdf_indirect=(alpha * aa(out[3].detach())).detach(), kd_grad=kd_grad * g["occ"], ks_grad=ks_grad * g["occ"],
               normal_grad=torch.zeros((N, 1), device=dev), occ=alpha, **extra)
    if ssaa > 1:      # every image goes through the SSAA down-scale before the losses see it (:1265-1297); T = 1 - alpha is scaled like the images
        res = {k: scale_img_hwc(v.view(fy, fx, v.shape[-1]), (H, W)).reshape(H * W, v.shape[-1]) for k, v in res.items()}
        fx, fy = W, H
    res["image_brdf"] = res["image_brdf"] + (1 - res["occ"]) * bg_color                                            # :1301
    return dict(res, fx=fx, fy=fy)                                                                                 # :1350-1352


def srgb_to_linear(x):
    """nerf/utils.py:57-58: the inverse tone curve the data loader applies to the training images (`images_linear`, :927) for the shading loss."""
    return torch.where(x < 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def linear2srgb(x):
    return torch.where(x <= 0.0031308, 12.92 * x, 1.055 * torch.pow(x + 1e-6, 1.0 / 2.4) - 0.055)


def scale_img_hwc(x, size):
    """nerf/renderer.py:61-76 for the case the harness uses (minification): plain bilinear resampling, align_corners False, NO antialiasing — for the
    exact 2x down-scale of --ssaa 2 that is the 2 x 2 box average."""
    assert x.shape[0] >= size[0] and x.shape[1] >= size[1]
    if x.shape[0] == size[0] and x.shape[1] == size[1]:
        return x
    return torch.nn.functional.interpolate(x.permute(2, 0, 1)[None], size, mode="bilinear")[0].permute(1, 2, 0).contiguous()


def postprocess(final_color, occ, H, W, ssaa):
    """clamp[0,1] -> sRGB -> x alpha -> SSAA down-scale (scale_img_hwc) -> + (1-alpha) * 1  (nerf/renderer.py:1125-1129,1162-1164,1208-1209,1265-1267,1301-1302)."""
    h, w = H * ssaa, W * ssaa
    img = linear2srgb(torch.clamp(final_color, 0.0, 1.0)).view(h, w, 3)
    alpha = (occ.view(h, w, 1) > 0.5).float()
    img = img * alpha
    if ssaa > 1:
        img = scale_img_hwc(img, (H, W))
        alpha = scale_img_hwc(alpha, (H, W))
    return img + (1 - alpha) * 1.0


def psnr(pred, gt):
    mse = torch.mean((pred - gt) ** 2)
    return float(-10.0 * torch.log10(mse))


def read_hdr(path):
    """Radiance RGBE (.hdr) reader -> float32 [H,W,3] RGB, for `--envmap_path` relighting (the reference uses cv2.imread(..., IMREAD_ANYDEPTH),
    nerf/network.py:136). Supports flat and new-style RLE scanlines."""
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    if not data.startswith(b"#?"):
        raise ValueError("not a Radiance HDR file")
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end]
        pos = end + 1
        if line == b"":
            break
    end = data.index(b"\n", pos)
    dims = data[pos:end].split()
    pos = end + 1
    if len(dims) != 4 or dims[0] != b"-Y" or dims[2] != b"+X":
        raise ValueError("unsupported orientation %r" % dims)
    H, W = int(dims[1]), int(dims[3])
    buf = np.frombuffer(data, dtype=np.uint8, offset=pos)
    rgbe = np.zeros((H, W, 4), np.uint8)
    p = 0
    for y in range(H):
        if W >= 8 and W < 32768 and buf[p] == 2 and buf[p + 1] == 2 and (int(buf[p + 2]) << 8 | int(buf[p + 3])) == W:
            p += 4
            for c in range(4):
                x = 0
                while x < W:
                    n = int(buf[p]); p += 1
                    if n > 128:
                        n -= 128
                        rgbe[y, x:x + n, c] = buf[p]; p += 1
                    else:
                        rgbe[y, x:x + n, c] = buf[p:p + n]; p += n
                    x += n
        else:
            rgbe[y] = buf[p:p + 4 * W].reshape(W, 4); p += 4 * W
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)
    return np.ascontiguousarray(rgbe[..., :3].astype(np.float32) * scale[..., None])


def write_hdr(path, img):
    """Flat (non-RLE) Radiance RGBE writer, used by the tests to make fixtures."""
    img = np.asarray(img, np.float32)
    H, W, _ = img.shape
    m = img.max(axis=2)
    mant, ex = np.frexp(m)
    sc = np.where(m > 1e-32, mant * 256.0 / np.maximum(m, 1e-32), 0.0)
    rgbe = np.zeros((H, W, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * sc[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, ex + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (H, W))
        f.write(rgbe.tobytes())
