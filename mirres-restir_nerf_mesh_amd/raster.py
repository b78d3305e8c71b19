"""The step in front of the path: what nerf/renderer.py:978-998 gets from nvdiffrast and meshutils — primary visibility, attribute
interpolation with gradients to the vertex attributes, smooth vertex normals — on the engine's own BVH (csrc/raster.hip).

    rast = rasterize_raycast(worker, rays_o, rays_d)          # stands in for dr.rasterize: [n,4] = (u, v, t, triangle_id + 1)
    xyzs = interpolate(vertices, rast, triangles)             # dr.interpolate(vertices, rast, triangles)[0]
    v_nrm, t_nrm_idx = auto_normals(vertices, triangles)      # meshutils.py:14-39
    gb_normal = interpolate(v_nrm, rast, t_nrm_idx)

    nrm_jitter = texture(gb_normal.view(h, w, 3), jitter)     # dr.texture(..., filter_mode='linear', boundary_mode='clamp') (:1004, :1008)

dr.antialias (the visibility gradient, :1184-1206) is not provided."""
import torch

from ._lib import lib, check, stream_ptr


@torch.no_grad()
def rasterize_raycast(worker, rays_o, rays_d):
    """Primary rays through worker's BVH (closest hit). Returns rast f32[n,4] in nvdiffrast's layout; the triangle id is exact in fp32 below 2^24 triangles."""
    n = rays_o.shape[0]
    rays = torch.empty((n, 8), dtype=torch.float32, device=rays_o.device)
    rays[:, 0:3] = rays_o; rays[:, 3] = 0.0; rays[:, 4:7] = rays_d; rays[:, 7] = 1e7
    rast = torch.zeros((n, 4), dtype=torch.float32, device=rays.device)
    vert = worker.vrt.detach().float().contiguous(); tri = worker.v_ind.detach().to(torch.int32).contiguous()
    check(lib().mirres_raster_raycast(worker.h, rays.data_ptr(), n, vert.data_ptr(), tri.data_ptr(), rast.data_ptr(), stream_ptr()), "mirres_raster_raycast")
    return rast


class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri):
        attr_c = attr.detach().float().contiguous(); rast_c = rast.detach().float().contiguous(); tri_c = tri.detach().to(torch.int32).contiguous()
        n, C = rast_c.shape[0], attr_c.shape[1]
        out = torch.empty((n, C), dtype=torch.float32, device=attr_c.device)
        check(lib().mirres_interpolate(attr_c.data_ptr(), C, rast_c.data_ptr(), tri_c.data_ptr(), n, out.data_ptr(), stream_ptr()), "mirres_interpolate")
        ctx.save_for_backward(attr_c, rast_c, tri_c)
        return out

    @staticmethod
    def backward(ctx, g_out):
        attr, rast, tri = ctx.saved_tensors
        g_out = g_out.contiguous().float()
        n, C = rast.shape[0], attr.shape[1]
        g_attr = torch.zeros_like(attr) if ctx.needs_input_grad[0] else None
        g_uv = torch.empty((n, 2), dtype=torch.float32, device=attr.device) if ctx.needs_input_grad[1] else None
        if g_attr is not None or g_uv is not None:
            check(lib().mirres_interpolate_bwd(attr.data_ptr(), C, rast.data_ptr(), tri.data_ptr(), n, g_out.data_ptr(), g_attr.data_ptr() if g_attr is not None else None,
                                               g_uv.data_ptr() if g_uv is not None else None, stream_ptr()), "mirres_interpolate_bwd")
        g_rast = None
        if g_uv is not None:
            g_rast = torch.zeros_like(rast); g_rast[:, 0:2] = g_uv
        return g_attr, g_rast, None


def interpolate(attr, rast, tri):
    """dr.interpolate(attr[None], rast, tri)[0] for one image flattened to [n,4] / [n,C]: attr f32[V,C] -> f32[n,C]; zeros where rast's triangle id is 0."""
    return _Interpolate.apply(attr, rast, tri)


def auto_normals(v_pos, t_pos_idx):
    """meshutils.py:14-39: area-weighted vertex normals — every face's (unnormalised) normal added to its three vertices, then normalised; a vertex
    whose sum is (numerically) zero gets (0, 0, 1). Returns (v_nrm [V,3], t_pos_idx) like the reference. Differentiable w.r.t. v_pos."""
    idx = t_pos_idx.to(torch.int64)
    corners = v_pos[idx]                                              # [T,3,3]
    fn = torch.cross(corners[:, 1] - corners[:, 0], corners[:, 2] - corners[:, 0], dim=-1)
    acc = torch.zeros_like(v_pos).index_add_(0, idx[:, 0], fn).index_add_(0, idx[:, 1], fn).index_add_(0, idx[:, 2], fn)
    sq = (acc * acc).sum(-1, keepdim=True)
    up = torch.zeros(3, dtype=v_pos.dtype, device=v_pos.device); up[2] = 1.0
    acc = torch.where(sq > 1e-20, acc, up)
    return acc / (acc * acc).sum(-1, keepdim=True).clamp(min=1e-20).sqrt(), t_pos_idx


class _Texture(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv):
        tex_c = tex.detach().float().contiguous(); uv_c = uv.detach().float().contiguous().view(-1, 2)
        H, W, C_ = tex_c.shape; n = uv_c.shape[0]
        out = torch.empty((n, C_), dtype=torch.float32, device=tex_c.device)
        check(lib().mirres_texture2d(tex_c.data_ptr(), H, W, C_, uv_c.data_ptr(), n, out.data_ptr(), stream_ptr()), "mirres_texture2d")
        ctx.save_for_backward(uv_c); ctx.dims = (H, W, C_)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (uv,) = ctx.saved_tensors; H, W, C_ = ctx.dims
        g_tex = None
        if ctx.needs_input_grad[0]:
            g_tex = torch.zeros((H, W, C_), dtype=torch.float32, device=uv.device)
            check(lib().mirres_texture2d_bwd(H, W, C_, uv.data_ptr(), uv.shape[0], g_out.contiguous().float().data_ptr(), g_tex.data_ptr(), stream_ptr()), "mirres_texture2d_bwd")
        return g_tex, None


def texture(tex, uv):
    """dr.texture(tex[None], uv[None], filter_mode='linear', boundary_mode='clamp')[0]: tex f32[H,W,C], uv f32[...,2] in [0,1] -> f32[prod(...), C];
    gradients flow to tex (the taps' coordinates are constants in the reference: jitter = pixel_grid + noise)."""
    return _Texture.apply(tex, uv)
