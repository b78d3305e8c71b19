"""Stage-1 loss and optimiser glue: what `Trainer.train_step` adds around `render_stage1` when `--stage 1 --use_brdf` trains the
material field, the environment map and the vertex offsets (nerf/utils.py:1003-1126, 1565-1589).

Host logic in torch (the reference's is torch too); the renderer underneath is the HIP path.  Every function cites the reference lines it stands in
for; `tests/golden/gen_reference_losses.py` runs the reference's own functions on seeded inputs and `tests/test_losses.py` compares values and
gradients.  Two regularisers of the reference come from pytorch3d (`mesh_normal_consistency`, `mesh_edge_loss`; `--lambda_normal` and
`--lambda_edgelen` default to 0, main.py:89-90): pytorch3d is not part of the reference tree, so those two restate its published definitions and
are checked on closed-form cases only.
"""
import torch

from .harness import linear2srgb

__all__ = ["shading_loss", "material_smoothness_grad", "material_extra_kd_smoothness_grad", "laplacian_smooth_loss", "mesh_edge_loss",
           "mesh_normal_consistency", "offsets_loss", "stage1_loss", "stage1_optimizer_step"]


def _third(x):
    return (x[..., 0] + x[..., 1] + x[..., 2]) / 3


def _srgb_unclipped(x):
    """nerf/utils.py:53-54 `linear_to_srgb`: exponent 0.41666, no clipping, no epsilon (the reference tone-maps the target with this one and the
    prediction with linear2srgb_torch — kept as is)."""
    return torch.where(x < 0.0031308, 12.92 * x, 1.055 * x ** 0.41666 - 0.055)


def shading_loss(diffuse_light, specular_light, color_ref, lambda_diffuse, lambda_specular):
    """Monochrome shading regulariser (nerf/utils.py:306-318).  The reference repeats each luma to three equal channels before its means; the mean of
    three equal columns is the mean of one, so this works on one column."""
    d = _third(diffuse_light)
    s = _third(specular_light)
    ref = color_ref[..., 0:3].amax(dim=-1)
    both = d + s
    img = linear2srgb(torch.clamp(torch.log(torch.clamp(both, 0, 65535) + 1), 0.0, 1.0))   # linear2srgb_torch clips its argument (utils.py:95)
    target = _srgb_unclipped(torch.log(torch.clamp(ref, 0, 65535) + 1))
    err = torch.abs(img - target) * d / torch.clamp(both, min=1e-3)
    return err.mean() * lambda_diffuse + s.mean() / torch.clamp(d.mean(), min=1e-3) * lambda_specular


def material_smoothness_grad(kd_grad, ks_grad, nrm_grad, lambda_kd=0.25, lambda_ks=0.1, lambda_nrm=0.0):
    """nerf/utils.py:277-282: means of the jittered-tap differences render_stage1 returns."""
    return _third(kd_grad).mean() * lambda_kd + ks_grad.mean() * lambda_ks + nrm_grad.mean() * lambda_nrm


def material_extra_kd_smoothness_grad(kd_grad, normal_ao, lambda_kd=0.25):
    """nerf/utils.py:284-288."""
    return (_third(kd_grad) * normal_ao[..., 0]).mean() * lambda_kd


def unique_edges(faces, n_verts):
    """Undirected edges of a triangle list, each once: [E, 2] int64 with e[:,0] < e[:,1]."""
    f = faces.long()
    a = torch.cat([f[:, 0], f[:, 1], f[:, 2]])
    b = torch.cat([f[:, 1], f[:, 2], f[:, 0]])
    lo, hi = torch.minimum(a, b), torch.maximum(a, b)
    keep = lo != hi
    key = torch.unique(lo[keep] * n_verts + hi[keep])
    return torch.stack([key // n_verts, key % n_verts], dim=1)


def laplacian_smooth_loss(verts, faces, cotan=False):
    """Uniform-Laplacian smoothness (nerf/utils.py:231-274 with cotan=False, the only form its call site uses, :1087): mean over vertices of
    |deg(i) v_i - sum_{j in N(i)} v_j|.  The reference assembles the sparse matrix D - A from the de-duplicated adjacency and multiplies; here the
    same sum is two index_adds over the unique edge list."""
    if cotan:
        raise NotImplementedError("the reference's call site never asks for the cotangent form")
    V = verts.shape[0]
    e = unique_edges(faces, V)
    i, j = e[:, 0], e[:, 1]
    diff = verts[i] - verts[j]                       # contributes +diff at i, -diff at j
    lv = torch.zeros_like(verts).index_add(0, i, diff).index_add(0, j, -diff)
    return lv.norm(dim=1).mean()


def mesh_edge_loss(verts, faces, target_length=0.0):
    """pytorch3d.loss.mesh_edge_loss for one mesh (used at nerf/utils.py:1101-1106): mean over the unique edges of (|e| - target)^2."""
    e = unique_edges(faces, verts.shape[0])
    return (((verts[e[:, 0]] - verts[e[:, 1]]).norm(dim=1) - target_length) ** 2).mean()


def mesh_normal_consistency(verts, faces):
    """pytorch3d.loss.mesh_normal_consistency for one mesh (used at nerf/utils.py:1094-1099): for every pair of faces that share an edge (v0, v1),
    with a and b the vertices opposite the edge, 1 - cos between (v1 - v0) x (a - v0) and -(v1 - v0) x (b - v0), averaged over the pairs.  An edge
    with more than two faces contributes all its pairs."""
    f = faces.long()
    V = verts.shape[0]
    a = torch.cat([f[:, 0], f[:, 1], f[:, 2]]); b = torch.cat([f[:, 1], f[:, 2], f[:, 0]]); o = torch.cat([f[:, 2], f[:, 0], f[:, 1]])
    lo, hi = torch.minimum(a, b), torch.maximum(a, b)
    key = lo * V + hi
    order = torch.argsort(key, stable=True)
    key, lo, hi, o = key[order], lo[order], hi[order], o[order]
    _, counts = torch.unique_consecutive(key, return_counts=True)
    if int(counts.max()) < 2:
        return verts.sum() * 0.0
    start = torch.cumsum(counts, 0) - counts
    pa, pb = [], []
    for c in torch.unique(counts).tolist():          # groups of c half-edges on one edge: all c (c - 1) / 2 pairs
        if c < 2:
            continue
        base = start[counts == c]
        for u in range(c):
            for w in range(u + 1, c):
                pa.append(base + u); pb.append(base + w)
    pa, pb = torch.cat(pa), torch.cat(pb)
    v0, v1 = verts[lo[pa]], verts[hi[pa]]
    n0 = torch.cross(v1 - v0, verts[o[pa]] - v0, dim=1)
    n1 = -torch.cross(v1 - v0, verts[o[pb]] - v0, dim=1)
    return (1 - torch.cosine_similarity(n0, n1, dim=1)).mean()


def offsets_loss(voffsets, n_inner=None):
    """L2 on the vertex offsets (nerf/utils.py:1108-1124): sum over xyz, mean over vertices; with --bound > 1 the outer mesh (from `n_inner` on)
    counts a tenth."""
    if n_inner is None:
        return (voffsets.abs() ** 2).sum(-1).mean()
    return (voffsets[:n_inner].abs() ** 2).sum(-1).mean() + 0.1 * (voffsets[n_inner:].abs() ** 2).sum(-1).mean()


def stage1_loss(outputs, gt_rgb, gt_rgb_linear, opt, vertices=None, voffsets=None, triangles=None, n_inner=None, criterion_lpips=None, frame_hw=None):
    """The stage-1 scalar of Trainer.train_step (nerf/utils.py:1003-1017, 1043-1126) from render_stage1's outputs.

    `opt` carries the reference's --lambda_* fields (main.py:81-113) and use_brdf; missing fields take main.py's defaults.  The perceptual term
    (--lambda_lpips > 0, default 0; utils.py:1079-1082) needs `criterion_lpips` (meters.LPIPS with the user's pretrained weights; the reference calls it
    on the [0, 1] images without `normalize`, kept) and the frame's (H, W); mask and refine-error terms need the reference's data loader and are outside
    this path."""
    g = lambda k, d: getattr(opt, k, d)
    use_brdf = g("use_brdf", True)
    loss = 0.0
    if "image" in outputs:     # the NeRF colour branch (stage 0's product); harness.render_stage1_outputs does not produce it
        loss = g("lambda_rgb", 1.0) * ((outputs["image"] - gt_rgb) ** 2).mean(-1)               # criterion = MSELoss(reduction='none'), main.py:231
    if use_brdf:
        loss = loss + g("lambda_rgb_brdf", 0.02) * (outputs["image_brdf"] - gt_rgb).abs().mean(-1)   # criterion_brdf = L1Loss, utils.py:788
    if not torch.is_tensor(loss):
        raise ValueError("stage1_loss: neither `image` nor (use_brdf and `image_brdf`) in outputs")
    loss = loss.mean()
    if use_brdf:
        loss = loss + shading_loss(outputs["diffuse_light"], outputs["specular_light"], gt_rgb_linear - outputs["img_brdf_indirect"],
                                   g("lambda_brdf_diffuse", 0.0015), g("lambda_brdf_specular", 0.000025))
        loss = loss + material_smoothness_grad(outputs["kd_grad"], outputs["ks_grad"], outputs["normal_grad"], lambda_kd=g("lambda_kd", 0.005),
                                               lambda_ks=g("lambda_ks", 0.0025), lambda_nrm=g("lambda_nrm", 0.00025))
        if g("lambda_extra_kd", 0.0) > 0:
            loss = loss + material_extra_kd_smoothness_grad(outputs["kd_grad"], outputs["normal_ao"].reshape(outputs["kd_grad"].shape), g("lambda_extra_kd", 0.0))
    if g("lambda_lpips", 0.0) > 0:
        if criterion_lpips is None or frame_hw is None:
            raise ValueError("stage1_loss: lambda_lpips > 0 needs criterion_lpips (meters.LPIPS with pretrained weights) and frame_hw")
        H, W = frame_hw
        img4 = lambda x: x.view(1, H, W, 3).permute(0, 3, 1, 2).contiguous()
        if "image" in outputs:
            loss = loss + g("lambda_lpips", 0.0) * criterion_lpips(img4(outputs["image"]), img4(gt_rgb)).reshape(())
        if use_brdf:
            loss = loss + g("lambda_lpips", 0.0) * criterion_lpips(img4(outputs["image_brdf"]), img4(gt_rgb)).reshape(())
    if vertices is not None and voffsets is not None:
        moved = vertices + voffsets                                                                 # act_voffsets is the identity (utils.py:341-347)
        if g("lambda_lap", 0.001) > 0:
            loss = loss + g("lambda_lap", 0.001) * laplacian_smooth_loss(moved, triangles)
        if g("lambda_normal", 0.0) > 0:
            loss = loss + g("lambda_normal", 0.0) * mesh_normal_consistency(moved, triangles)
        if g("lambda_edgelen", 0.0) > 0:
            loss = loss + g("lambda_edgelen", 0.0) * mesh_edge_loss(moved, triangles)
        if g("lambda_offsets", 0.1) > 0:
            loss = loss + g("lambda_offsets", 0.1) * offsets_loss(voffsets, n_inner)
    return loss


def stage1_optimizer_step(loss, optimizer, optimizer_mat=None, optimizer_light=None, light_base=None, encoder_params=None, scheduler=None,
                          scheduler_mat=None, scheduler_light=None, grad_sync=None):
    """backward + the three optimiser steps of nerf/utils.py:1565-1589: geometry first, then the environment-map gradient x 64 and the hash-grid
    gradient / 8, material and light steps, and the light clamped at 0.01 from below.  (The reference wraps the geometry step in a GradScaler; fp16
    is off on this path, where the scaler is the identity.)  The caller zeroes the three optimisers' gradients before rendering, as :1555-1558 does.
    `grad_sync` (data-parallel training: one view or strip per rank) is called between the backward pass and the first optimiser step — e.g.
    `lambda: dist.allreduce_gradients([voffsets, *mlp.parameters(), light_base])`, one flat bucket of every parameter gradient (all-to-all of slices + rank-ordered sum + all-gather over RCCL; dist.allreduce_gradients)."""
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    if optimizer_mat is not None or optimizer_light is not None:
        if light_base is not None and light_base.grad is not None:
            light_base.grad *= 64
        if encoder_params is not None and encoder_params.grad is not None:
            encoder_params.grad /= 8.0
        if optimizer_mat is not None:
            optimizer_mat.step()
            if scheduler_mat is not None:
                scheduler_mat.step()
        if optimizer_light is not None:
            optimizer_light.step()
            if scheduler_light is not None:
                scheduler_light.step()
        if light_base is not None:
            with torch.no_grad():
                light_base.clamp_(min=0.01)
    return float(loss.detach())
