"""Multi-GPU sharding of one frame. One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU tests).

Scheme used here: the spp range of the frame is split into contiguous slices, one per rank (disjoint frame-index ranges, so every rank
draws the samples a single GPU would have drawn for those indices); mesh/BVH/env/material are replicated. The ONLY data-path exchange is
one all-reduce(sum) of the six [N,3] accumulators after the loop, followed by the (cheap, replicated) average + EAW + composite.
Temporal reuse restarts at the first sample of a slice, i.e. the result is statistically equivalent, not bit-identical, to 1 GPU."""
import ctypes as C

import torch


def spp_slice(spp, rank, world):
    """Contiguous [begin, end) of sample indices for `rank`; sizes differ by at most one; empty slices allowed when spp < world."""
    base, rem = divmod(int(spp), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_sums(sums, group=None):
    """Sum the six accumulators over ranks as ONE flat collective (184 MB at 1600x1600: large enough to be link-bandwidth bound)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sums
    flat = torch.cat([s.reshape(-1) for s in sums])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, o = [], 0
    for s in sums:
        n = s.numel()
        out.append(flat[o:o + n].view_as(s))
        o += n
    return out


def render_sharded(ctx, worker, mlp_mat, env_map, g, spp, random_offset, rank, world, denoise_iter=2, step_width=2, c_phi=2.0, n_phi=0.1, p_phi=0.001,
                   use_scale=False, scale=(1.0, 1.0, 1.0), group=None):
    """Renders this rank's spp slice with the fused loop, all-reduces the raw sums and finishes (average, EAW, composite) on every rank.
    `g` is a G-buffer dict (harness.build_gbuffer). Returns the six [N,3] outputs of run_restir_di_with_pt."""
    from . import _lib
    from ._lib import lib, check, stream_ptr
    from .renderer_restir import render_fused
    b, e = spp_slice(spp, rank, world)
    occ = g["occ"].clone()
    if world == 1:
        outs, _, _ = render_fused(ctx, worker, mlp_mat, use_scale, scale, env_map, occ, g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp,
                                  denoise_iter, step_width, c_phi, n_phi, p_phi, random_offset)
        return outs
    # an empty slice (more ranks than samples) has begin == end > 0: the loop body never runs and the sums stay zero
    sums, a, keep = render_fused(ctx, worker, mlp_mat, use_scale, scale, env_map, occ, g["normal"], g["depth"], g["kd"], g["rm"], g["ray_dir"], g["pos"], spp,
                                 denoise_iter, step_width, c_phi, n_phi, p_phi, random_offset, spp_range=(b, e))
    sums = allreduce_sums(sums, group)
    outs = [torch.empty_like(s) for s in sums]
    for k in range(6):
        a.outs[k] = outs[k].data_ptr()
    arr = (C.c_void_p * 6)(*[s.data_ptr() for s in sums])
    check(lib().mirres_render_finish(ctx.h, C.byref(a), arr, stream_ptr()), "mirres_render_finish")
    return outs
