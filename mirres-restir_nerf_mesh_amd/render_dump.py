"""Mirror of nerf/render_dump.py (reference): the direct-lighting renderer used by render_stage1 when --use_restir is off
(nerf/renderer.py:1131-1149; BASELINE configs[0]). Same names and argument meaning; the work is done by libmirres.so (csrc/dump.hip).

The reference takes its `intersector` from outside (nerf/renderer.py:179 leaves it None); here it is a `restirbvhWorker` (renderer_restir.py),
whose BVH answers the occlusion queries — `batch_intersector` works with it as with any object offering `intersects_closest`."""
import torch

from ._lib import lib, check, stream_ptr


def safe_l2_normalize(x, dim=None, eps=1e-6):
    """render_dump.py:5-6."""
    return torch.nn.functional.normalize(x, p=2, dim=dim, eps=eps)


@torch.no_grad()
def batch_intersector(intersector, rays_o, rays_d, vis_near, chunk_size):
    """render_dump.py:8-27: visibility [N,1] (0 where the offset ray hits something). One query for the whole batch (the reference's chunks of
    `chunk_size` only bound its intermediate tensors)."""
    hit = intersector.intersects_closest(rays_o + rays_d * vis_near, rays_d, stream_compaction=True)[0]
    vis = torch.ones(rays_o.shape[0], dtype=torch.float32, device=rays_o.device)
    vis[hit] = 0.0
    return vis.reshape(-1, 1)


def _f32(x):
    return x.detach().to(dtype=torch.float32).contiguous()


def _run(intersector, surface_xyz, normal_map, albedo_map, roughness_map, fresnel_map, rays_d, env_map, env_h, env_w, model, sample_method, clamp_rgb):
    if intersector is None or not hasattr(intersector, "h"):
        raise ValueError("dump_render needs a restirbvhWorker as `intersector` (the reference leaves it to the caller, nerf/renderer.py:179)")
    dev = surface_xyz.device
    n = surface_xyz.shape[0]
    dirs = _f32(model.fixed_viewdirs.to(dev)).reshape(-1, 3); L = dirs.shape[0]
    w = _f32(model.light_area_weight.to(dev)).reshape(-1)
    env = _f32(env_map.to(dev)).reshape(env_h, env_w, 3)
    a = [_f32(t.to(dev)).reshape(n, 3) for t in (surface_xyz, normal_map, albedo_map, roughness_map, fresnel_map, rays_d)]
    lrgb = torch.empty((L, 3), dtype=torch.float32, device=dev)
    outs = [torch.zeros((n, 3), dtype=torch.float32, device=dev) for _ in range(3)]
    check(lib().mirres_dump_render(intersector.h, n, L, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(), a[5].data_ptr(),
                                   env.data_ptr(), int(env_h), int(env_w), dirs.data_ptr(), w.data_ptr(), int(sample_method == 'stratifed_sample_equal_areas'),
                                   int(clamp_rgb), lrgb.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), stream_ptr()), "mirres_dump_render")
    return outs[0], outs[1], outs[2]


def dump_render_run_mesh(intersector, surface_xyz, normal_map, albedo_map, roughness_map, fresnel_map, rays_d, env_map, env_h, env_w, model,
                         sample_method='stratified_sampling', chunk_size=15000, device='cuda', use_linear2srgb=True):
    """render_dump.py:136-215 -> (rgb_with_brdf, rgb_with_brdf_diff, rgb_with_brdf_spec), unclamped."""
    return _run(intersector, surface_xyz, normal_map, albedo_map, roughness_map, fresnel_map, rays_d, env_map, env_h, env_w, model, sample_method, False)


def dump_render(intersector, surface_xyz, normal_map, albedo_map, roughness_map, fresnel_map, rays_d, env_map, env_h, env_w, model,
                sample_method='stratified_sampling', color_chunk_size=15000, chunk_size=15000, device='cuda', use_linear2srgb=True):
    """render_dump.py:84-133 -> (brdf_color clamped to [0,1], diff_color, spec_color), each [N,3]. The chunk sizes only bounded the reference's
    [chunk, lights, 3] temporaries; the engine chunks internally."""
    return _run(intersector, surface_xyz, normal_map, albedo_map, roughness_map, fresnel_map, rays_d, env_map, env_h, env_w, model, sample_method, True)
