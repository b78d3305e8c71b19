"""nerf/renderutils/ops.py:164-211 — bilateral_denoiser / bilateral_denoiser_no_di on the HIP kernels (mirres_bilateral, mirres_bilateral_bwd).
Same names, arguments and return values as the reference; the kernels replace the `renderutils_plugin` CUDA extension (c_src/denoising.cu).
The normal is normalised inside the kernel's packing step (safe_normalize, :168-169), so no gradient flows to it — as in the reference,
whose autograd Function returns None for nrm and zdz (:181-185)."""
import torch

from .._lib import lib, check, stream_ptr, MirresError


def _f32c(t):
    if not t.is_cuda or t.dtype != torch.float32:
        raise MirresError("bilateral_denoiser expects CUDA float32 tensors, got %s on %s" % (t.dtype, t.device))
    return t.contiguous()


class _bilateral_denoiser_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, col, nrm, zdz, sigma, h, w):
        col, nrm, zdz = _f32c(col.detach()), _f32c(nrm.detach()), _f32c(zdz.detach())
        n = h * w
        out = torch.empty((n, 4), dtype=torch.float32, device=col.device)
        scratch = torch.empty((n, 8), dtype=torch.float32, device=col.device)
        check(lib().mirres_bilateral(int(w), int(h), float(sigma), col.data_ptr(), nrm.data_ptr(), zdz.data_ptr(), scratch.data_ptr(), out.data_ptr(), stream_ptr()),
              "mirres_bilateral")
        ctx.save_for_backward(nrm, zdz)
        ctx.sigma, ctx.h, ctx.w = float(sigma), int(h), int(w)
        return out

    @staticmethod
    def backward(ctx, out_grad):
        nrm, zdz = ctx.saved_tensors
        n = ctx.h * ctx.w
        g = _f32c(out_grad)
        col_grad = torch.empty((n, 3), dtype=torch.float32, device=g.device)
        scratch = torch.empty((n, 8), dtype=torch.float32, device=g.device)
        check(lib().mirres_bilateral_bwd(ctx.w, ctx.h, ctx.sigma, nrm.data_ptr(), zdz.data_ptr(), g.data_ptr(), scratch.data_ptr(), col_grad.data_ptr(), stream_ptr()),
              "mirres_bilateral_bwd")
        return col_grad, None, None, None, None, None


def bilateral_denoiser(h, w, input, factor=1.0):
    """input f32[h*w, 8] = (colour, normal, z, |dz|) -> f32[h*w, 3]; differentiable w.r.t. the colour (ops.py:190-199)."""
    if input.shape[-1] != 8:
        raise ValueError("bilateral_denoiser: input must have 8 channels (colour 3, normal 3, z, |dz|), got %d" % input.shape[-1])
    input = input.reshape(h * w, input.shape[-1])
    sigma = max(factor * 2, 0.0001)
    col_w = _bilateral_denoiser_func.apply(input[:, 0:3], input[:, 3:6], input[:, 6:8], sigma, h, w)
    out_val = col_w[..., 0:3] / col_w[..., 3:4]
    return out_val.view(-1, out_val.shape[-1])


@torch.no_grad()
def bilateral_denoiser_no_di(h, w, input, factor=1.0):
    """The same filter without autograd bookkeeping (ops.py:201-211)."""
    return bilateral_denoiser(h, w, input, factor)


# ---------------------------------------------------------------------------------------------- prepare_shading_normal (ops.py:100-163)
class _prepare_shading_normal_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl):
        shape = torch.broadcast_shapes(pos.shape, view_pos.shape, perturbed_nrm.shape, smooth_nrm.shape, smooth_tng.shape, geom_nrm.shape)
        ins = [_f32c(t.detach().expand(shape)) for t in (pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm)]
        n = ins[0].numel() // 3
        out = torch.empty(shape, dtype=torch.float32, device=ins[0].device)
        check(lib().mirres_prepare_shading_normal(n, *[t.data_ptr() for t in ins], int(bool(two_sided_shading)), int(bool(opengl)), out.data_ptr(), stream_ptr()),
              "mirres_prepare_shading_normal")
        ctx.save_for_backward(*ins)
        ctx.flags = (int(bool(two_sided_shading)), int(bool(opengl)))
        ctx.in_shapes = [tuple(t.shape) for t in (pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm)]
        return out

    @staticmethod
    def backward(ctx, dout):
        ins = ctx.saved_tensors
        n = ins[0].numel() // 3
        g = [torch.empty_like(ins[0]) if need else None for need in ctx.needs_input_grad[:6]]
        p = lambda t: t.data_ptr() if t is not None else None
        check(lib().mirres_prepare_shading_normal_bwd(n, *[t.data_ptr() for t in ins], ctx.flags[0], ctx.flags[1], _f32c(dout.expand(ins[0].shape)).data_ptr(),
                                                      *[p(t) for t in g], stream_ptr()), "mirres_prepare_shading_normal_bwd")
        # inputs that were broadcast (view_pos [1,1,1,3], the default perturbed normal) get their gradient summed back to their own shape
        return tuple(t.sum_to_size(s) if t is not None else None for t, s in zip(g, ctx.in_shapes)) + (None, None)


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True, use_python=False):
    """ops.py:128-163: final shading normal (tangent-space perturbation, two-sided flip, bending of back-facing normals). `use_python` is accepted
    for signature compatibility; there is one implementation (the HIP kernels)."""
    if perturbed_nrm is None:
        perturbed_nrm = torch.tensor([0, 0, 1], dtype=torch.float32, device=pos.device, requires_grad=False)[None, None, None, ...]
    out = _prepare_shading_normal_func.apply(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl)
    if torch.is_anomaly_enabled():
        assert torch.all(torch.isfinite(out)), "Output of prepare_shading_normal contains inf or NaN"
    return out
