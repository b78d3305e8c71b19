"""Mirror of nerf/ScreenSpaceReSTIR/Denoising.py (reference): edge-avoiding a-trous denoiser on the MI355X engine."""
import torch

from ._lib import lib, check, stream_ptr
from ._ops import _f32


def _eaw(c_phi, n_phi, p_phi, fx, fy, step, occ_map, color, normal_map, pos_map):
    out = torch.empty((int(fx) * int(fy), 3), dtype=torch.float32, device=color.device)
    check(lib().mirres_eaw(int(fx), int(fy), int(step), float(c_phi), float(n_phi), float(p_phi), _f32(occ_map).data_ptr(), _f32(color).data_ptr(),
                           _f32(normal_map).data_ptr(), _f32(pos_map).data_ptr(), out.data_ptr(), stream_ptr()), "mirres_eaw")
    return out


class EAWDenoise_run(torch.autograd.Function):
    """Denoising.py:10-48."""

    @staticmethod
    def forward(ctx, m, c_phi, n_phi, p_phi, framedim_x, framedim_y, stepWidth, occ_map, color, normal_map, pos_map):
        occ_map, color, normal_map, pos_map = (_f32(t.detach()) for t in (occ_map, color, normal_map, pos_map))
        out_color = _eaw(c_phi, n_phi, p_phi, framedim_x, framedim_y, stepWidth, occ_map, color, normal_map, pos_map)
        ctx.save_for_backward(occ_map, color, normal_map, pos_map)
        ctx.nums = [c_phi, n_phi, p_phi, framedim_x, framedim_y, stepWidth]
        return out_color

    @staticmethod
    def backward(ctx, grad_out_color):
        grad_out_color = grad_out_color.contiguous()
        occ_map, color, normal_map, pos_map = ctx.saved_tensors
        c_phi, n_phi, p_phi, fx, fy, step = ctx.nums
        grad_color = torch.zeros_like(color)
        grad_normal = torch.zeros_like(normal_map)
        grad_pos = torch.zeros_like(pos_map)
        scratch = torch.empty((color.shape[0], 4), dtype=torch.float32, device=color.device)
        check(lib().mirres_eaw_bwd_gather(int(fx), int(fy), int(step), float(c_phi), float(n_phi), float(p_phi), occ_map.data_ptr(), color.data_ptr(),
                                          normal_map.data_ptr(), pos_map.data_ptr(), grad_out_color.data_ptr(), scratch.data_ptr(), grad_color.data_ptr(),
                                          grad_normal.data_ptr(), grad_pos.data_ptr(), stream_ptr()), "mirres_eaw_bwd_gather")
        return (None, None, None, None, None, None, None, None, grad_color, grad_normal, grad_pos)


def EAWDenoise_run_no_di(m, c_phi, n_phi, p_phi, framedim_x, framedim_y, stepWidth, occ_map, color, normal_map, pos_map):
    """Denoising.py:50-60."""
    return _eaw(c_phi, n_phi, p_phi, framedim_x, framedim_y, stepWidth, occ_map, color, normal_map, pos_map)


def EAWDenoise_use_phi(m, c_phi, n_phi, p_phi, stepWidth, iter_time, framedim_x, framedim_y, occ_map, color, normal_map, pos_map):
    """Denoising.py:154-203: iter_time a-trous passes with step stepWidth, stepWidth/2, ..."""
    curr_color = color
    out_color = color
    for _ in range(iter_time):
        out_color = EAWDenoise_run.apply(m, c_phi, n_phi, p_phi, framedim_x, framedim_y, stepWidth, occ_map, curr_color, normal_map, pos_map)
        curr_color = out_color
        stepWidth /= 2
    return out_color


@torch.no_grad()
def EAWDenoise_use_phi_no_di(m, c_phi, n_phi, p_phi, stepWidth, iter_time, framedim_x, framedim_y, occ_map, color, normal_map, pos_map):
    """Denoising.py:205-251."""
    curr_color = color
    out_color = color
    for _ in range(iter_time):
        out_color = EAWDenoise_run_no_di(m, c_phi, n_phi, p_phi, framedim_x, framedim_y, stepWidth, occ_map, curr_color.detach(), normal_map.detach(),
                                         pos_map.detach())
        curr_color = out_color
        stepWidth /= 2
    return out_color
