"""The step in front of the path: what nerf/renderer.py:978-998 gets from nvdiffrast and meshutils — primary visibility, attribute
interpolation with gradients to the vertex attributes, smooth vertex normals — on the engine's own BVH (csrc/raster.hip).

    rast = rasterize_raycast(worker, rays_o, rays_d)          # stands in for dr.rasterize: [n,4] = (u, v, t, triangle_id + 1)
    xyzs = interpolate(vertices, rast, triangles)             # dr.interpolate(vertices, rast, triangles)[0]
    v_nrm, t_nrm_idx = auto_normals(vertices, triangles)      # meshutils.py:14-39
    gb_normal = interpolate(v_nrm, rast, t_nrm_idx)

    nrm_jitter = texture(gb_normal.view(h, w, 3), jitter)     # dr.texture(..., filter_mode='linear', boundary_mode='clamp') (:1004, :1008)

    topo = antialias_topology(triangles)                      # once per index buffer
    img = antialias(img.view(1, h, w, 3), rast.view(1, h, w, 4), vertices_clip, triangles, topology_hash=topo, pos_gradient_boost=...)   # dr.antialias (:1184-1206)"""
import ctypes as C

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


class RasterizeContext:
    """What the reference creates as `dr.RasterizeGLContext()` / `dr.RasterizeCudaContext()` (nerf/renderer.py:186-190) and passes to dr.rasterize as
    `glctx`: here it names the restirbvhWorker whose BVH (world space, rebuilt by update_mesh at :975 just before) the primary rays are cast through."""
    def __init__(self, worker=None):
        self.worker = worker


@torch.no_grad()
def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True, mvp=None):
    """dr.rasterize(glctx, pos, tri, resolution) (nerf/renderer.py:983): pos f32[1,V,4] (or [V,4]) clip-space vertices = pad(vertices, 1) @ mvp^T, tri
    i32[T,3], resolution (h, w) -> (rast [1,h,w,4], rast_db [1,h,w,4]) in nvdiffrast's layout: (u, v, z/w, triangle_id + 1) with perspective-correct
    barycentrics, and (du/dX, du/dY, dv/dX, dv/dY).  `glctx` = RasterizeContext(worker) whose worker holds the same mesh in world space (worker.vrt);
    the model-view-projection matrix is `mvp` [4,4] if given, else recovered from (worker.vrt, pos) — pos is an exact linear image of the vertices, so
    a float64 least-squares fit returns the matrix to rounding.  Not differentiable (neither is nvdiffrast's: gradients enter through dr.interpolate
    and dr.antialias).  Clipping: every pixel's ray starts on the near plane (z_c = -w_c) and takes the nearest triangle in front of that point
    (mirres_bvh_trace mode 4), so a triangle crossing the near plane shows its far part and hides nothing with its near part; beyond the far plane: empty.
    rast_db is analytic (csrc/raster.hip)."""
    if ranges is not None:
        raise NotImplementedError("rasterize: range mode (instanced minibatches) is not used by the path")
    worker = glctx.worker if isinstance(glctx, RasterizeContext) else glctx
    if worker is None or not hasattr(worker, "vrt"):
        raise ValueError("rasterize: glctx must be a RasterizeContext holding the restirbvhWorker of this mesh")
    p = pos.detach().reshape(-1, 4)
    vert = worker.vrt.detach().float().contiguous(); t32 = tri.detach().to(torch.int32).contiguous()
    if p.shape[0] != vert.shape[0]:
        raise ValueError("rasterize: %d clip-space vertices for a BVH over %d vertices" % (p.shape[0], vert.shape[0]))
    h, w = int(resolution[0]), int(resolution[1])
    if mvp is None:
        A = torch.cat((vert, torch.ones_like(vert[:, :1])), dim=1).double()
        step = max(1, A.shape[0] // 4096)
        mvp = torch.linalg.lstsq(A[::step], p[::step].double()).solution.t()            # p = A @ mvp^T
        # the fit is only meaningful when `pos` really is a linear image of THIS worker's vertices (a BVH built over another mesh of the same vertex
        # count would otherwise rasterise silently wrong): the residual of an exact image is float32 rounding of the clip coordinates
        resid = float((A[::step] @ mvp.t() - p[::step].double()).abs().max()); scale = max(1.0, float(p[::step].abs().max()))
        if not resid <= 1e-4 * scale:
            raise ValueError("rasterize: clip-space vertices are not a projective image of the worker's mesh (fit residual %.3e): wrong BVH for `pos`?" % resid)
    M = mvp.detach().double().cpu()
    A = M[[0, 1, 3]]                                             # rows x, y, w: the eye is where all three vanish
    if abs(float(torch.linalg.det(A[:, :3]))) < 1e-12 * float(A[:, :3].abs().max()) ** 3:
        raise NotImplementedError("rasterize: the matrix is not a perspective projection (no eye point); the path's cameras are")
    eye = torch.linalg.solve(A[:, :3], -A[:, 3])
    cm = (C.c_float * 16)(*[float(x) for x in M.reshape(-1)]); cmi = (C.c_float * 3)(*[float(x) for x in eye])
    rast = torch.empty((h * w, 4), dtype=torch.float32, device=vert.device)
    rast_db = torch.empty((h * w, 4), dtype=torch.float32, device=vert.device) if grad_db else None
    check(lib().mirres_rasterize(worker.h, vert.data_ptr(), t32.data_ptr(), cm, cmi, w, h, rast.data_ptr(), rast_db.data_ptr() if grad_db else None, stream_ptr()), "mirres_rasterize")
    return rast.view(1, h, w, 4), (rast_db.view(1, h, w, 4) if grad_db else torch.zeros((1, h, w, 0), device=vert.device))


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


def dr_interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """dr.interpolate(attr, rast, tri, rast_db=None, diff_attrs=None) with nvdiffrast's shapes: attr [1,V,C] (or [V,C]), rast [1,h,w,4] -> (out [1,h,w,C],
    out_db).  out_db = image-space derivatives [1,h,w,2 C'] = (da/dX, da/dY) per selected attribute when `rast_db` is given AND `diff_attrs` selects
    attributes ('all' or a list of indices); with diff_attrs=None it is an EMPTY tensor, as in nvdiffrast — which is what nerf/renderer.py:1074 receives
    (it passes rast_db but no diff_attrs, and :1078-1079 then substitutes ones for the empty gradient)."""
    a = attr.reshape(-1, attr.shape[-1])
    shp = rast.shape[:-1]
    r = rast.reshape(-1, 4)
    out = interpolate(a, r, tri).view(*shp, a.shape[1])
    if rast_db is None or diff_attrs is None:
        return out, torch.zeros((*shp, 0), dtype=out.dtype, device=out.device)
    sel = list(range(a.shape[1])) if diff_attrs == "all" else [int(i) for i in diff_attrs]
    idx = (r[:, 3].long() - 1).clamp(min=0)
    t = tri.to(torch.int64)[idx]
    a0, a1, a2 = a[t[:, 0]][:, sel], a[t[:, 1]][:, sel], a[t[:, 2]][:, sel]
    db = rast_db.reshape(-1, 4)
    on = (r[:, 3:4] > 0).to(a.dtype)
    dX = (db[:, 0:1] * (a0 - a2) + db[:, 2:3] * (a1 - a2)) * on
    dY = (db[:, 1:2] * (a0 - a2) + db[:, 3:4] * (a1 - a2)) * on
    return out, torch.stack((dX, dY), dim=-1).reshape(*shp, 2 * len(sel))


def dr_texture(tex, uv, filter_mode="linear", boundary_mode="clamp"):
    """dr.texture(tex [1,H,W,C], uv [1,h,w,2], filter_mode='linear', boundary_mode='clamp') -> [1,h,w,C] (nerf/renderer.py:1003-1008)."""
    if filter_mode != "linear" or boundary_mode != "clamp":
        raise NotImplementedError("texture: the path uses filter_mode='linear', boundary_mode='clamp' only")
    t = tex.reshape(tex.shape[-3], tex.shape[-2], tex.shape[-1])
    return texture(t, uv).view(*uv.shape[:-1], t.shape[-1])


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


def antialias_topology(tri):
    """For every edge (v_k, v_k+1) of every triangle: the vertex opposite that edge in the neighbouring triangle, or -1 on a boundary edge — what
    dr.antialias needs to tell silhouette edges from interior ones (nvdiffrast's `antialias_construct_topology_hash`).  Depends on the index buffer
    only: built once per mesh (torch sort over the 3 T directed edges; set-up, not per-frame work).  An edge shared by more than two triangles pairs
    each triangle with the next one in sorted order.  Returns i32[T,3] on tri's device."""
    t = tri.detach().to(torch.int64)
    T = t.shape[0]
    a = torch.stack((t[:, 0], t[:, 1], t[:, 2]), 1).reshape(-1)            # edge k starts at v_k ...
    b = torch.stack((t[:, 1], t[:, 2], t[:, 0]), 1).reshape(-1)            # ... ends at v_k+1 ...
    c = torch.stack((t[:, 2], t[:, 0], t[:, 1]), 1).reshape(-1)            # ... and faces v_k+2
    lo, hi = torch.minimum(a, b), torch.maximum(a, b)
    key = lo * (int(t.max().item()) + 1 if T else 1) + hi
    order = torch.argsort(key, stable=True)
    ks = key[order]; cs = c[order]
    opp_sorted = torch.full_like(cs, -1)
    same_next = torch.zeros_like(ks, dtype=torch.bool); same_prev = torch.zeros_like(ks, dtype=torch.bool)
    if ks.numel() > 1:
        eq = ks[1:] == ks[:-1]
        same_next[:-1] = eq; same_prev[1:] = eq
    nxt = torch.roll(cs, -1); prv = torch.roll(cs, 1)
    opp_sorted = torch.where(same_next, nxt, torch.where(same_prev, prv, opp_sorted))
    opp = torch.empty_like(opp_sorted); opp[order] = opp_sorted
    return opp.view(T, 3).to(torch.int32).contiguous()


class _Antialias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, opp, H, W, boost):
        col = color.detach().float().contiguous(); r = rast.detach().float().contiguous(); p = pos.detach().float().contiguous()
        t = tri.detach().to(torch.int32).contiguous()
        C_ = col.shape[-1]
        out = torch.empty_like(col)
        check(lib().mirres_antialias(W, H, C_, col.data_ptr(), r.data_ptr(), p.data_ptr(), t.data_ptr(), opp.data_ptr(), out.data_ptr(), stream_ptr()), "mirres_antialias")
        ctx.save_for_backward(col, r, p, t, opp); ctx.dims = (H, W, C_, float(boost))
        return out

    @staticmethod
    def backward(ctx, g_out):
        col, r, p, t, opp = ctx.saved_tensors; H, W, C_, boost = ctx.dims
        g = g_out.contiguous().float()
        g_col = torch.empty_like(col) if ctx.needs_input_grad[0] else None
        g_pos = torch.zeros_like(p) if ctx.needs_input_grad[2] else None
        if g_col is not None or g_pos is not None:
            check(lib().mirres_antialias_bwd(W, H, C_, col.data_ptr(), r.data_ptr(), p.data_ptr(), t.data_ptr(), opp.data_ptr(), g.data_ptr(),
                                             g_col.data_ptr() if g_col is not None else None, g_pos.data_ptr() if g_pos is not None else None, boost, stream_ptr()),
                  "mirres_antialias_bwd")
        return g_col, None, g_pos, None, None, None, None, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """dr.antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0) for one image: color [1,H,W,C] (or [H,W,C]), rast [1,H,W,4]
    as rasterize_raycast returns it (reshaped), pos [1,V,4] / [V,4] clip-space vertex positions, tri [T,3].  `topology_hash` = antialias_topology(tri)
    (built on the fly when None, as nvdiffrast does).  Differentiable w.r.t. color and pos; returns the shape of `color`."""
    shape = color.shape
    if color.dim() == 4:
        if shape[0] != 1:
            raise ValueError("antialias: one image per call (minibatch of 1), got %s" % (tuple(shape),))
        H, W, C_ = shape[1:]
    elif color.dim() == 3:
        H, W, C_ = shape
    else:
        raise ValueError("antialias: color must be [1,H,W,C] or [H,W,C]")
    if rast.numel() != H * W * 4:
        raise ValueError("antialias: rast has %d values, expected %d x %d x 4" % (rast.numel(), H, W))
    p = pos.reshape(-1, 4)
    if topology_hash is None:
        topology_hash = antialias_topology(tri)
    if topology_hash.shape != tri.shape or topology_hash.dtype != torch.int32:
        raise ValueError("antialias: topology_hash must be antialias_topology(tri) (i32[T,3])")
    out = _Antialias.apply(color.reshape(H * W, C_), rast.reshape(H * W, 4), p, tri, topology_hash.contiguous(), int(H), int(W), float(pos_gradient_boost))
    return out.view(shape)


class _Dr:
    """`import nvdiffrast.torch as dr` of nerf/renderer.py, on the engine's own operators: dr.RasterizeCudaContext / RasterizeGLContext, dr.rasterize,
    dr.interpolate, dr.texture, dr.antialias with nvdiffrast's argument order and tensor shapes."""
    RasterizeCudaContext = RasterizeContext
    RasterizeGLContext = RasterizeContext
    rasterize = staticmethod(rasterize)
    interpolate = staticmethod(dr_interpolate)
    texture = staticmethod(dr_texture)
    antialias = staticmethod(antialias)


dr = _Dr()
