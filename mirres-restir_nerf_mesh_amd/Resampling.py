"""Mirror of nerf/ScreenSpaceReSTIR/Resampling.py (reference): launch wrappers + the two differentiable operators,
running on the MI355X engine. Functions that take the LBVH node arrays find the owning restirbvhWorker through them."""
import ctypes as C
import weakref
import torch

from . import _lib
from ._lib import lib, check, stream_ptr
from ._ops import _f32, env_struct, gbuf_struct, res_struct, path_struct

# LBVHNode_info.data_ptr() -> restirbvhWorker.  Weak: the registry must not keep a worker (and the device memory of its BVH) alive once the caller has
# dropped it — with a plain dict the worker's __del__ could never run.
_BVH_OWNERS = weakref.WeakValueDictionary()


def _owner(LBVHNode_info):
    w = _BVH_OWNERS.get(LBVHNode_info.data_ptr())
    if w is None:
        raise _lib.MirresError("LBVHNode_info does not belong to a live restirbvhWorker")
    return w


def _u32(v):
    return int(v) & 0xffffffff


def InitialResampling_(m, LBVHNode_info, LBVHNode_aabb, vert, vert_ind, pos_map, reservoirs, env_tex, env_width, env_height, framedim_x, framedim_y,
                       frameIndex, occ_map, normal_depth, brdf_map, ray_dir, pdf_, cdf_, mpdf_, mcdf_, light_data, light_uv, light_inv_pdf):
    """Resampling.py:7-25."""
    keep = []
    e = env_struct(env_tex, env_width, env_height, pdf_, cdf_, mpdf_, mcdf_, keep)
    g = gbuf_struct(occ_map, pos_map, normal_depth, brdf_map, ray_dir, keep)
    r = res_struct(reservoirs, keep)
    check(lib().mirres_restir_initial(m.ctx.h, _owner(LBVHNode_info).h, C.byref(e), C.byref(g), C.byref(r), _f32(light_data).data_ptr(),
                                      _f32(light_inv_pdf).data_ptr(), _u32(frameIndex), stream_ptr()), "mirres_restir_initial")
    return 'hello'


def TemporalResampling(m, reservoirs, prev_reservoirs, env_tex, env_width, env_height, framedim_x, framedim_y, frameIndex, occ_map, normal_depth, brdf_map,
                       ray_dir, prev_occ_map, prev_normal_depth, prev_brdf_map, prev_ray_dir, motionVectors):
    """Resampling.py:26-45."""
    keep = []
    e = env_struct(env_tex, env_width, env_height, None, None, None, None, keep)
    g = gbuf_struct(occ_map, None, normal_depth, brdf_map, ray_dir, keep)
    pg = gbuf_struct(prev_occ_map, None, prev_normal_depth, prev_brdf_map, prev_ray_dir, keep)
    r = res_struct(reservoirs, keep); pr = res_struct(prev_reservoirs, keep)
    mv = _f32(motionVectors) if motionVectors is not None else None
    check(lib().mirres_restir_temporal(m.ctx.h, C.byref(e), C.byref(g), C.byref(pg), C.byref(r), C.byref(pr), mv.data_ptr() if mv is not None else None,
                                       _u32(frameIndex), stream_ptr()), "mirres_restir_temporal")
    return 'hello'


def SpatialResampling_(m, LBVHNode_info, LBVHNode_aabb, vert, vert_ind, pos_map, reservoirs, prev_reservoirs, neighborOffsets, env_tex, env_width, env_height,
                       framedim_x, framedim_y, frameIndex, occ_map, normal_depth, brdf_map, ray_dir):
    """Resampling.py:47-66."""
    keep = []
    e = env_struct(env_tex, env_width, env_height, None, None, None, None, keep)
    g = gbuf_struct(occ_map, pos_map, normal_depth, brdf_map, ray_dir, keep)
    r = res_struct(reservoirs, keep); pr = res_struct(prev_reservoirs, keep)
    check(lib().mirres_restir_spatial(m.ctx.h, _owner(LBVHNode_info).h, C.byref(e), C.byref(g), C.byref(r), C.byref(pr), _f32(neighborOffsets).data_ptr(),
                                      _u32(frameIndex), stream_ptr()), "mirres_restir_spatial")
    return 'hello'


def EvaluateFinalSamples_get_vis(m, LBVHNode_info, LBVHNode_aabb, vert, vert_ind, pos_map, reservoirs, framedim_x, framedim_y, vis_map):
    """Resampling.py:80-92. Mutates vis_map in place."""
    keep = []
    r = res_struct(reservoirs, keep)
    check(lib().mirres_restir_final_vis(m.ctx.h, _owner(LBVHNode_info).h, _f32(pos_map).data_ptr(), C.byref(r), vis_map.data_ptr(), stream_ptr()),
          "mirres_restir_final_vis")
    return 'hello'


class EvaluateFinalSamples_di(torch.autograd.Function):
    """Resampling.py:94-143. Writes finalSamples_dir / finalSamples_distance in place, returns final_Li; differentiable w.r.t. env_tex.
    Unlike the reference the reservoir state needed by backward is snapshotted, so the gradient is the
    adjoint of the forward that actually ran for this sample."""

    @staticmethod
    def forward(ctx, m, res_light_data, res_light_pdf, res_M, res_weight, env_tex, env_width, env_height, framedim_x, framedim_y, finalSamples_dir,
                finalSamples_distance, eva_vis_map):
        keep = []
        env_d = _f32(env_tex.detach())
        e = env_struct(env_d, env_width, env_height, None, None, None, None, keep)
        r = res_struct((res_light_data, res_light_pdf, res_M, res_weight), keep)
        final_Li = torch.empty((int(framedim_x) * int(framedim_y), 3), dtype=torch.float32, device=env_d.device)
        check(lib().mirres_restir_eval_final(m.ctx.h, C.byref(e), C.byref(r), _f32(eva_vis_map).data_ptr(), finalSamples_dir.data_ptr(),
                                             finalSamples_distance.data_ptr(), final_Li.data_ptr(), stream_ptr()), "mirres_restir_eval_final")
        if env_tex.requires_grad:
            ctx.save_for_backward(res_light_data.clone(), res_light_pdf.clone(), res_M.clone(), res_weight.clone(), env_d, eva_vis_map.clone())
        ctx.nums = [env_width, env_height]
        ctx.m = m
        return final_Li

    @staticmethod
    def backward(ctx, grad_final_Li):
        grad_final_Li = grad_final_Li.contiguous()
        ld, pdf, M, w, env_d, vis = ctx.saved_tensors
        env_width, env_height = ctx.nums
        keep = []
        e = env_struct(env_d, env_width, env_height, None, None, None, None, keep)
        r = res_struct((ld, pdf, M, w), keep)
        grad_env = torch.zeros_like(env_d)
        check(lib().mirres_restir_eval_final_bwd(ctx.m.ctx.h, C.byref(e), C.byref(r), vis.data_ptr(), grad_final_Li.data_ptr(), grad_env.data_ptr(), stream_ptr()),
              "mirres_restir_eval_final_bwd")
        return (None, None, None, None, None, grad_env, None, None, None, None, None, None, None)


class FinalShading(torch.autograd.Function):
    """Resampling.py:145-214. Returns (color, color_diff, color_spec); differentiable w.r.t. finalSamples_Li, normal, diffuse_map and
    linearRoughness_specular_map."""

    @staticmethod
    def forward(ctx, m, finalSamples_dir, finalSamples_distance, finalSamples_Li, env_tex, env_width, env_height, framedim_x, framedim_y, occ_map, normal,
                ray_dir, diffuse_map, linearRoughness_specular_map):
        N = int(framedim_x) * int(framedim_y)
        keep = []
        e = env_struct(env_tex.detach(), env_width, env_height, None, None, None, None, keep)
        ins = [_f32(t.detach()) for t in (occ_map, normal, ray_dir, diffuse_map, linearRoughness_specular_map, finalSamples_dir, finalSamples_distance,
                                          finalSamples_Li)]
        color = torch.empty((N, 3), dtype=torch.float32, device=ins[0].device)
        color_diff = torch.empty_like(color); color_spec = torch.empty_like(color)
        check(lib().mirres_final_shading(m.ctx.h, C.byref(e), *[t.data_ptr() for t in ins], color.data_ptr(), color_diff.data_ptr(), color_spec.data_ptr(),
                                         stream_ptr()), "mirres_final_shading")
        needs = any(t.requires_grad for t in (finalSamples_Li, normal, diffuse_map, linearRoughness_specular_map))
        if needs:
            # finalSamples_dir / _distance are persistent buffers overwritten by later samples: snapshot (Appendix B.19)
            ctx.save_for_backward(ins[0], ins[1], ins[2], ins[3], ins[4], ins[5].clone(), ins[6].clone(), ins[7])
        ctx.m = m
        return color, color_diff, color_spec

    @staticmethod
    def backward(ctx, grad_color, grad_color_diff, grad_color_spec):
        occ, normal, ray_dir, kd, rm, fdir, fdist, fLi = ctx.saved_tensors
        gc, gd, gs = (g.contiguous() for g in (grad_color, grad_color_diff, grad_color_spec))
        g_normal = torch.empty_like(normal); g_kd = torch.empty_like(kd); g_rm = torch.empty_like(rm); g_Li = torch.empty_like(fLi)
        check(lib().mirres_final_shading_bwd(ctx.m.ctx.h, occ.data_ptr(), normal.data_ptr(), ray_dir.data_ptr(), kd.data_ptr(), rm.data_ptr(), fdir.data_ptr(),
                                             fdist.data_ptr(), fLi.data_ptr(), gc.data_ptr(), gd.data_ptr(), gs.data_ptr(), g_normal.data_ptr(), g_kd.data_ptr(),
                                             g_rm.data_ptr(), g_Li.data_ptr(), stream_ptr()), "mirres_final_shading_bwd")
        return (None, None, None, g_Li, None, None, None, None, None, None, g_normal, None, g_kd, g_rm)


def process_new_dir_for_pt(m, LBVHNode_info, LBVHNode_aabb, vert, vert_ind, frameIndex, bounce_count, framedim_x, framedim_y, occ_map, pos_map, normal, ray_dir,
                           prd, diffuse_map, linearRoughness_specular_map, new_pos_map, new_ray_d, new_occ_map, new_normal):
    """Resampling.py:216-232."""
    keep = []
    p = path_struct(occ_map, pos_map, normal, ray_dir, diffuse_map, linearRoughness_specular_map, prd, new_pos_map, new_ray_d, new_occ_map, new_normal, keep)
    check(lib().mirres_pt_new_dir(m.ctx.h, _owner(LBVHNode_info).h, C.byref(p), _u32(frameIndex), int(bounce_count), stream_ptr()), "mirres_pt_new_dir")
    return 'hello'


def indirect_one_hit_divided_no_grad(m, LBVHNode_info, LBVHNode_aabb, vert, vert_ind, frameIndex, bounce_count, framedim_x, framedim_y, env_tex, env_width,
                                     env_height, pdf_, cdf_, mpdf_, mcdf_, occ_map, pos_map, normal, ray_dir, prd, diffuse_map, linearRoughness_specular_map,
                                     color, diff_color, spec_color, new_pos_map, new_ray_d, new_occ_map, new_normal):
    """Resampling.py:254-273. Overwrites color / diff_color / spec_color and the next-vertex buffers in place."""
    keep = []
    e = env_struct(env_tex, env_width, env_height, pdf_, cdf_, mpdf_, mcdf_, keep)
    p = path_struct(occ_map, pos_map, normal, ray_dir, diffuse_map, linearRoughness_specular_map, prd, new_pos_map, new_ray_d, new_occ_map, new_normal, keep)
    check(lib().mirres_pt_bounce(m.ctx.h, _owner(LBVHNode_info).h, C.byref(e), C.byref(p), _u32(frameIndex), int(bounce_count), color.data_ptr(),
                                 diff_color.data_ptr(), spec_color.data_ptr(), stream_ptr()), "mirres_pt_bounce")
    return 'hello'
