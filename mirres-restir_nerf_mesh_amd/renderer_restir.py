"""Mirror of nerf/renderer_restir.py (reference): restirbvhWorker, load_m_for_restir, restir_di_with_pt, run_restir_di_with_pt.

Same names, argument order, in-place mutation and return structure as the reference so that NeRFRenderer.render_stage1
(nerf/renderer.py:975,1113-1123) can call this module unchanged. Two execution paths produce the same numbers:
  * fused   : one C call (mirres_render) enqueues the whole spp loop — used when no input requires grad;
  * stepwise: the reference's per-pass Python loop over the same kernels — used under autograd (stage-1 training)."""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, stream_ptr
from ._ops import Module, get_ctx, _f32
from .Denoising import EAWDenoise_use_phi, EAWDenoise_use_phi_no_di
from .renderutils.ops import bilateral_denoiser, bilateral_denoiser_no_di   # renderer_restir.py:11
from .GenerateLightTiles import make_sampleable, GenerateLightTiles
from . import Resampling
from .Resampling import (TemporalResampling, EvaluateFinalSamples_di, FinalShading, process_new_dir_for_pt, indirect_one_hit_divided_no_grad)

_FIXED_RANDOM_OFFSET = None


def set_random_offset(v):
    """Pin the per-frame seed the reference draws with np.random.randint(2**20) (renderer_restir.py:245); None restores the draw."""
    global _FIXED_RANDOM_OFFSET
    _FIXED_RANDOM_OFFSET = v


def safe_l2_normalize(x, dim=None, eps=1e-6):
    return torch.nn.functional.normalize(x, p=2, dim=dim, eps=eps)


class restirbvhWorker:
    """renderer_restir.py:13-146: owns the LBVH; `.vrt`, `.v_ind`, `.LBVHNode_info`, `.LBVHNode_aabb` are read by callers."""

    def __init__(self, vt, vt_ind):
        self.vrt = vt
        self.v_ind = vt_ind
        self.h = None
        self._cap = 0
        self.LBVHNode_info = None
        self.LBVHNode_aabb = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _release(self):
        if self.LBVHNode_info is not None:
            Resampling._BVH_OWNERS.pop(self.LBVHNode_info.data_ptr(), None)
        if self.h:
            lib().mirres_bvh_destroy(self.h)
            self.h = None

    def _ensure(self, T):
        if self.h is None or T > self._cap:
            if self.h:
                torch.cuda.synchronize()
                lib().mirres_bvh_destroy(self.h)
            h = C.c_void_p()
            check(lib().mirres_bvh_create(C.byref(h), int(T)), "mirres_bvh_create")
            self.h, self._cap = h, int(T)

    def update_bvh(self):
        """renderer_restir.py:25-89 -> (LBVHNode_info i32[2T-1,3], LBVHNode_aabb f32[2T-1,6]); no host synchronisation."""
        vrt = _f32(self.vrt.detach())
        v_ind = self.v_ind
        if v_ind.dtype != torch.int32:
            v_ind = v_ind.to(torch.int32)
        v_ind = v_ind.contiguous()
        T = v_ind.shape[0]
        self._ensure(T)
        info = torch.empty((2 * T - 1, 3), dtype=torch.int32, device=vrt.device)
        aabb = torch.empty((2 * T - 1, 6), dtype=torch.float32, device=vrt.device)
        # The private steering hierarchy is built in two steps (round 6): the extended-Morton tree now, its binned-SAH top (0.9-1.4 ms per build) only when a frame is
        # long enough to pay for it (ensure_hierarchy_for, called by render_fused). MIRRES_PRIVATE_TREE = 0 / 1 / 2 fixes the level at build time as before.
        env = os.environ.get("MIRRES_PRIVATE_TREE")
        level = int(env) if env not in (None, "", "auto") else 1
        check(lib().mirres_bvh_build_level(self.h, vrt.data_ptr(), vrt.shape[0], v_ind.data_ptr(), T, info.data_ptr(), aabb.data_ptr(), None, level, stream_ptr()),
              "mirres_bvh_build_level")
        self._keep = (vrt, v_ind)
        self._auto_level = env in (None, "", "auto")
        return info, aabb

    SAH_TOP_FROM_PIXEL_SAMPLES = 1.0e8      # the SAH top returns ~4-7 % of a frame's traversal time: from about 1e8 pixel-samples on that is more than its build (800 x 800 x 32 spp = 2e7: not; 1600 x 1600 x 512 = 1.3e9: yes; profiles/r06_ab_train_tree.txt)

    def upgrade(self):
        """Completes the private hierarchy (binned-SAH top) on top of the current build; a no-op when it is complete or was fixed by MIRRES_PRIVATE_TREE."""
        if self.h is None or self.LBVHNode_info is None:
            return
        vrt, v_ind = self._keep
        check(lib().mirres_bvh_upgrade(self.h, vrt.data_ptr(), v_ind.data_ptr(), self.LBVHNode_info.data_ptr(), self.LBVHNode_aabb.data_ptr(), stream_ptr()), "mirres_bvh_upgrade")

    def ensure_hierarchy_for(self, pixel_samples):
        if getattr(self, "_auto_level", False) and pixel_samples >= self.SAH_TOP_FROM_PIXEL_SAMPLES:
            self.upgrade()

    def update_mesh(self, vt, vt_ind):
        if self.LBVHNode_info is not None:
            Resampling._BVH_OWNERS.pop(self.LBVHNode_info.data_ptr(), None)
        self.vrt = vt
        self.v_ind = vt_ind
        self.LBVHNode_info, self.LBVHNode_aabb = self.update_bvh()
        Resampling._BVH_OWNERS[self.LBVHNode_info.data_ptr()] = self

    def trace(self, rays_o, rays_d, closest=True, t_min=0.0, t_max=1e7):
        """bvh_hit / bvh_hit_with_normal over a ray batch (helperDi.slang:197-395). Returns dict(hit, t, pos, normal, prim)."""
        n = rays_o.shape[0]
        rays = torch.empty((n, 8), dtype=torch.float32, device=rays_o.device)
        rays[:, 0:3] = rays_o; rays[:, 3] = t_min; rays[:, 4:7] = rays_d; rays[:, 7] = t_max
        hit = torch.empty(n, dtype=torch.int32, device=rays.device)
        if not closest:
            check(lib().mirres_bvh_trace(self.h, rays.data_ptr(), n, 0, hit.data_ptr(), None, None, None, None, None, stream_ptr()), "mirres_bvh_trace")
            return dict(hit=hit)
        t = torch.empty(n, dtype=torch.float32, device=rays.device); pos = torch.empty((n, 3), dtype=torch.float32, device=rays.device)
        nrm = torch.empty_like(pos); prim = torch.empty(n, dtype=torch.int32, device=rays.device)
        check(lib().mirres_bvh_trace(self.h, rays.data_ptr(), n, 1, hit.data_ptr(), t.data_ptr(), pos.data_ptr(), nrm.data_ptr(), prim.data_ptr(), None, stream_ptr()),
              "mirres_bvh_trace")
        return dict(hit=hit, t=t, pos=pos, normal=nrm, prim=prim)

    def intersects_closest(self, rays_o, rays_d, stream_compaction=True):
        """What nerf/render_dump.py:batch_intersector (:8-27) asks of its external `intersector`: (hit mask, ...) of a conventional ray tracer
        (hits in front of the origin only — mirres_bvh_trace mode 3). The other five return values of the reference's intersector are unused there."""
        n = rays_o.shape[0]
        rays = torch.empty((n, 8), dtype=torch.float32, device=rays_o.device)
        rays[:, 0:3] = rays_o; rays[:, 3] = 0.0; rays[:, 4:7] = rays_d; rays[:, 7] = 1e7
        hit = torch.zeros(n, dtype=torch.int32, device=rays.device)
        if n:
            check(lib().mirres_bvh_trace(self.h, rays.data_ptr(), n, 3, hit.data_ptr(), None, None, None, None, None, stream_ptr()), "mirres_bvh_trace")
        return hit > 0, None, None, None, None, None

    def InitialResampling_(self, m, pos_map, reservoirs, env_tex, env_width, env_height, framedim_x, framedim_y, frameIndex, occ_map, normal_depth, brdf_map,
                           ray_dir, pdf_, cdf_, mpdf_, mcdf_, light_data, light_uv, light_inv_pdf):
        return Resampling.InitialResampling_(m, self.LBVHNode_info, self.LBVHNode_aabb, self.vrt, self.v_ind, pos_map, reservoirs, env_tex, env_width, env_height,
                                             framedim_x, framedim_y, frameIndex, occ_map, normal_depth, brdf_map, ray_dir, pdf_, cdf_, mpdf_, mcdf_, light_data,
                                             light_uv, light_inv_pdf)

    def SpatialResampling_(self, m, pos_map, reservoirs, prev_reservoirs, neighborOffsets, env_tex, env_width, env_height, framedim_x, framedim_y, frameIndex,
                           occ_map, normal_depth, brdf_map, ray_dir):
        return Resampling.SpatialResampling_(m, self.LBVHNode_info, self.LBVHNode_aabb, self.vrt, self.v_ind, pos_map, reservoirs, prev_reservoirs,
                                             neighborOffsets, env_tex, env_width, env_height, framedim_x, framedim_y, frameIndex, occ_map, normal_depth, brdf_map,
                                             ray_dir)

    def EvaluateFinalSamples_get_vis(self, m, pos_map, reservoirs, framedim_x, framedim_y, vis_map):
        return Resampling.EvaluateFinalSamples_get_vis(m, self.LBVHNode_info, self.LBVHNode_aabb, self.vrt, self.v_ind, pos_map, reservoirs, framedim_x,
                                                       framedim_y, vis_map)


def load_m_for_restir(framedim_x, framedim_y):
    """renderer_restir.py:148-228 -> the same 17-tuple (8 module handles, light tiles, reservoirs, final samples, neighbour offsets, tile shape)."""
    ctx = get_ctx(framedim_x, framedim_y)
    names = ("make_sampleable", "generateLightTiles", "InitialResampling", "TemporalResampling", "SpatialResampling", "EvaluateFinalSamples", "FinalShading",
             "denoising")
    mods = tuple(Module(n, ctx) for n in names)
    light_tile_count, light_tile_size = ctx.cfg.light_tile_count, ctx.cfg.light_tile_size
    N = int(framedim_x) * int(framedim_y)
    dev = 'cuda'
    light_data = torch.zeros((light_tile_count * light_tile_size, 3), dtype=torch.float, device=dev)
    light_uv = torch.zeros((light_tile_count * light_tile_size, 2), dtype=torch.int, device=dev)
    light_inv_pdf = torch.zeros((light_tile_count * light_tile_size, 1), dtype=torch.float, device=dev)

    def _res():
        return (torch.zeros((N, 3), dtype=torch.float, device=dev), torch.zeros((N, 1), dtype=torch.float, device=dev),
                torch.zeros((N, 1), dtype=torch.int, device=dev), torch.zeros((N, 1), dtype=torch.float, device=dev))
    reservoirs = _res()
    final_samples = (torch.zeros((N, 3), dtype=torch.float, device=dev), torch.zeros((N, 1), dtype=torch.float, device=dev),
                     torch.zeros((N, 3), dtype=torch.float, device=dev))
    prev_reservoirs = _res()
    start_time = time.time()
    neighborOffsets = torch.zeros((ctx.cfg.neighbor_offset_count, 2), dtype=torch.float, device=dev)
    check(lib().mirres_neighbor_offsets(ctx.h, neighborOffsets.data_ptr(), stream_ptr()), "mirres_neighbor_offsets")
    print(f"Create neighbor offset time consumed: {time.time() - start_time} s")
    return mods + (light_data, light_uv, light_inv_pdf, reservoirs, prev_reservoirs, final_samples, neighborOffsets, light_tile_count, light_tile_size)


class _PassClock:
    """Frame-index bookkeeping of the sample loop.  Every pass seeds its per-pixel RNG streams with `offset + STRIDE * sample + tick`, where the tick advances
    by a fixed amount after each pass (renderer_restir.py:314-456: tiles +2, initial +1, temporal +1 — skipped for the first sample, which is why the spatial
    pass of sample 0 runs one tick earlier —, spatial +1, new direction +5, every indirect vertex +5) and STRIDE = 5 + 15 ticks separate two samples (:316)."""
    STRIDE = 5 + 15
    TICKS = {"tiles": 2, "initial": 1, "temporal": 1, "spatial": 1, "direct": 0, "new_dir": 5, "vertex": 5}

    def __init__(self, offset):
        self.offset, self.sample, self.tick = int(offset), 0, 0

    @property
    def index(self):
        return self.offset + self.STRIDE * self.sample + self.tick

    def done(self, stage):
        self.tick += self.TICKS[stage]

    def next_sample(self):
        self.sample += 1
        self.tick = 0


def _target_function_inputs(normal_map, depth_map, diffuse_map, roughness_specular):
    """What the reservoir passes see of the material (renderer_restir.py:279-287): (n, depth) and the three scalars of the target function — diffuse weight =
    luminance of kd, specular weight = `metallic` weighted by the three luminance coefficients (they sum to one), GGX alpha = clamp(roughness, 0.01, 1)^2."""
    lum = diffuse_map.new_tensor([0.2126, 0.7152, 0.0722])
    wd = diffuse_map[:, 0:1] * lum[0] + diffuse_map[:, 1:2] * lum[1] + diffuse_map[:, 2:3] * lum[2]
    m = roughness_specular[:, 1:2]
    ws = m * lum[0] + m * lum[1] + m * lum[2]
    alpha = roughness_specular[:, 0:1].clamp(min=0.01, max=1)
    return torch.cat((normal_map, depth_map), dim=-1).detach(), torch.cat((wd, ws, alpha * alpha), dim=-1).detach().contiguous()


def restir_di_with_pt(use_scale, scale_x, scale_y, scale_z, mlp_mat, bvh_restir_worker, spp, framedim_x, framedim_y, make_sampleable_m, generateLightTiles_m,
                      InitialResampling_m, TemporalResampling_m, SpatialResampling_m, EvaluateFinalSamples_m, FinalShading_m, light_data, light_uv, light_inv_pdf,
                      reservoirs, prev_reservoirs, final_samples, neighborOffsets, light_tile_count, light_tile_size, env_map_init, occ_map, pos_map, normal_map,
                      depth_map, diffuse_map, roughness_specular, ray_dir_map, prev_occ_map, prev_normal_depth, prev_brdf_map, prev_ray_dir, motionVectors, color):
    """renderer_restir.py:230-471, the sample-by-sample path (signature and return tuple are the reference's; the one-call path is render_fused).  Used when a
    caller asks for the reference-shaped loop under autograd (MIRRES_TRAIN_FUSED=0): EvaluateFinalSamples_di and FinalShading are autograd Functions exactly
    where the reference has them; everything else runs on detached inputs.  The caller's prev_* arguments are ignored and history starts empty, as in the
    reference (:291-302)."""
    fx, fy = int(framedim_x), int(framedim_y)
    N = fx * fy
    dev = occ_map.device
    buf = lambda c: torch.zeros((N, c), dtype=torch.float32, device=dev)
    W = bvh_restir_worker
    mesh = lambda: (W.LBVHNode_info, W.LBVHNode_aabb, W.vrt, W.v_ind)
    geo, target = _target_function_inputs(normal_map, depth_map, diffuse_map, roughness_specular)
    # the environment as the kernels index it: rows flipped, flattened; the differentiable copy feeds EvaluateFinalSamples_di only
    Hc, Wc = env_map_init.shape[0], env_map_init.shape[1]
    env_grad = torch.flip(env_map_init, dims=[0]).reshape(-1, env_map_init.shape[2])
    env = env_grad.detach().contiguous()
    tables = make_sampleable(make_sampleable_m, env, Wc, Hc)
    # two reservoir sets with fixed roles: `fresh` receives a sample's initial candidates and its merge with the history; `merged` receives the spatial pass's
    # output, is what the sample is shaded with, and is the next sample's history (the reference reaches the same hand-over with two swaps per sample, :358 / :460)
    fresh = reservoirs
    merged = (buf(3), buf(1), torch.zeros((N, 1), dtype=torch.int32, device=dev), buf(1))            # empty history
    hist_gbuf = (torch.zeros_like(occ_map), buf(4), torch.zeros_like(target), torch.zeros_like(ray_dir_map))
    visible = torch.ones((N, 1), dtype=torch.float32, device=dev)
    # sums over samples: direct (colour, diffuse, specular) and indirect (colour, diffuse, specular); the seventh output of the reference stays zero (:298)
    direct = [buf(3), buf(3), buf(3)]
    indirect = [buf(3), buf(3), buf(3)]
    throughput = buf(5)
    kd_v, rm_v = buf(3), buf(2)                                                                       # material at the current path vertex
    vertex = [dict(occ=buf(1), pos=buf(3), normal=buf(3), dir=buf(3)) for _ in range(2)]               # ping-pong of path vertices
    contrib = [buf(3), buf(3), buf(3)]
    bounces = InitialResampling_m.ctx.cfg.max_bounce
    clock = _PassClock(np.random.randint(2**20) if _FIXED_RANDOM_OFFSET is None else int(_FIXED_RANDOM_OFFSET))
    for s_ in range(int(spp)):
        GenerateLightTiles(generateLightTiles_m, None, env, *tables, Wc, Hc, clock.index, light_data, light_uv, light_inv_pdf,
                           light_tile_count=light_tile_count, light_tile_size=light_tile_size)
        clock.done("tiles")
        W.InitialResampling_(InitialResampling_m, pos_map, fresh, env, Wc, Hc, fx, fy, clock.index, occ_map, geo, target, ray_dir_map, *tables,
                             light_data, light_uv, light_inv_pdf)
        clock.done("initial")
        if s_ > 0:
            TemporalResampling(TemporalResampling_m, fresh, merged, env, Wc, Hc, fx, fy, clock.index, occ_map, geo, target, ray_dir_map, *hist_gbuf, motionVectors)
            clock.done("temporal")
        W.SpatialResampling_(SpatialResampling_m, pos_map, merged, fresh, neighborOffsets, env, Wc, Hc, fx, fy, clock.index, occ_map, geo, target, ray_dir_map)
        clock.done("spatial")
        # direct lighting of the selected sample: visibility, W * Le, shading (the two differentiable stages)
        W.EvaluateFinalSamples_get_vis(EvaluateFinalSamples_m, pos_map, merged, fx, fy, visible)
        Li = EvaluateFinalSamples_di.apply(EvaluateFinalSamples_m, merged[0], merged[1], merged[2], merged[3], env_grad, Wc, Hc, fx, fy,
                                           final_samples[0], final_samples[1], visible)
        shaded = FinalShading.apply(FinalShading_m, final_samples[0], final_samples[1], Li, env, Wc, Hc, fx, fy, occ_map, normal_map, ray_dir_map, diffuse_map,
                                    roughness_specular)
        for k in range(3):
            direct[k] = direct[k] + shaded[k]
        # indirect lighting: a direction at the primary hit, then `bounces` path vertices with the material field looked up at each
        here, there = vertex
        process_new_dir_for_pt(FinalShading_m, *mesh(), clock.index, 0, fx, fy, occ_map, pos_map, normal_map.detach(), ray_dir_map, throughput,
                               diffuse_map.detach(), roughness_specular.detach(), here["pos"], here["dir"], here["occ"], here["normal"])
        clock.done("new_dir")
        for depth in range(1, bounces + 1):         # the reference writes depth = 1, 2 out by hand (:396-454)
            hit = torch.where(here["occ"] >= 0.5)[0]
            looked_up = mlp_mat.sample_no_di(here["pos"][hit])
            kd_v[hit] = looked_up[..., 0:3]
            rm_v[hit] = looked_up[..., 4:6]
            if use_scale:                           # relighting: albedo scaled per channel, then the WHOLE map clamped (:404-408)
                kd_v[hit] = kd_v[hit] * kd_v.new_tensor([scale_x, scale_y, scale_z])
                kd_v = torch.clamp(kd_v, min=0.0, max=1.0)
            indirect_one_hit_divided_no_grad(FinalShading_m, *mesh(), clock.index, depth, fx, fy, env, Wc, Hc, *tables, here["occ"], here["pos"], here["normal"],
                                             here["dir"], throughput, kd_v, rm_v, contrib[0], contrib[1], contrib[2], there["pos"], there["dir"], there["occ"],
                                             there["normal"])
            for k in range(3):
                indirect[k] += contrib[k]
            clock.done("vertex")
            here, there = there, here
        hist_gbuf = (occ_map, geo, target, ray_dir_map)      # from the second sample on the history's G-buffer is the frame's own
        clock.next_sample()
    return direct[0], indirect[0], direct[1], direct[2], indirect[1], indirect[2], buf(3), clock.sample


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def render_fused(ctx, bvh_restir_worker, mlp_mat, use_scale, scale, env_map, occ_map, normal_map, depth_map, diffuse_map, roughness_specular, ray_dir_map, pos_map,
                 spp, denoise_iter, stepWidth, c_phi, n_phi, p_phi, random_offset, spp_range=None, const_kd=(0.6, 0.6, 0.6), const_rm=(0.5, 0.0),
                 strip=None, halo=None, gb_depth=None, tape=None, strip_overlap=False, halo_native=None):
    """One C call for the whole frame (mirres_render). Returns the 6 output buffers [N,3] (raw sums when spp_range or strip is given).
    strip = (full_fy, y_off, own_y0, own_y1): `ctx` and all per-pixel inputs describe a rank's LOCAL frame (own rows + halo rows, dist.py);
    halo = a _lib.HALO_FN called once per sample to exchange the halo rows of the packed reservoirs; halo_native = (communicator handle of mirres_comm_create,
    plan [(peer, (send rows), (recv rows)), ...] in local rows, time stride): the library issues the exchange itself (RCCL send / receive group, no callback)."""
    N = ctx.N
    a = _lib.RenderArgs()
    keep = []
    a.spp, a.random_offset = int(spp), int(random_offset) & 0xffffffff
    a.use_scale = int(bool(use_scale)); a.scale[:] = [float(s) for s in scale]
    env = _f32(env_map.detach()); keep.append(env)
    a.env_map, a.Hc, a.Wc = env.data_ptr(), env.shape[0], env.shape[1]
    if not occ_map.is_contiguous():
        raise _lib.MirresError("occ_map is modified in place and must be contiguous")
    if env.dim() != 3 or env.shape[2] != 3:
        raise _lib.MirresError("env_map must be [Hc, Wc, 3], got %s" % (tuple(env.shape),))
    if occ_map.dtype != torch.float32 or occ_map.numel() != N:
        raise _lib.MirresError("occ_map must be float32 with %d elements (framedim_x * framedim_y), got %s %s" % (N, occ_map.dtype, tuple(occ_map.shape)))
    a.occ = occ_map.data_ptr(); keep.append(occ_map)     # thresholded in place; mirres_render_bwd / mirres_render_finish read it again
    # the kernels index these as [N, width] rows: a tensor of another size would be read out of bounds, so it is refused here
    for name, t, width in (("normal", normal_map, 3), ("depth", depth_map, 1), ("kd", diffuse_map, 3), ("rough_metal", roughness_specular, 2), ("ray_dir", ray_dir_map, 3),
                           ("pos", pos_map, 3)):
        if t.numel() != N * width:
            raise _lib.MirresError("%s must have %d x %d elements, got %s" % (name, N, width, tuple(t.shape)))
        t = _f32(t.detach()); keep.append(t); setattr(a, name, t.data_ptr())
    if mlp_mat is not None:
        st = mlp_mat._struct(); keep.append(st)
        a.mat = C.pointer(st)
    else:
        a.mat = None
    a.const_kd[:] = list(const_kd); a.const_rm[:] = list(const_rm)
    a.denoise_iter, a.step_width, a.c_phi, a.n_phi, a.p_phi = int(denoise_iter), int(stepWidth), float(c_phi), float(n_phi), float(p_phi)
    outs = [torch.empty((N, 3), dtype=torch.float32, device=env.device) for _ in range(6)]
    for k in range(6):
        a.outs[k] = outs[k].data_ptr()
    if tape is not None:          # per-sample record for mirres_render_bwd (training)
        keep.append(tape); a.tape = tape.data_ptr()
    if gb_depth is not None:      # bilateral denoiser instead of EAW (--use_bi_de)
        gd = _f32(gb_depth.detach()); keep.append(gd); a.gb_depth = gd.data_ptr()
    if spp_range is not None:
        a.spp_begin, a.spp_end = int(spp_range[0]), int(spp_range[1])
    if strip is not None:
        a.strip_full_fy, a.strip_y_off, a.own_y0, a.own_y1 = (int(v) for v in strip)
        if halo_native is not None:
            comm, plan, stride = halo_native
            if len(plan) > 2:
                raise _lib.MirresError("a strip has at most two neighbouring ranks")
            a.halo_comm = int(comm); a.halo_n = len(plan); a.halo_time_stride = int(stride)
            for k, (peer, (sa, sb), (ra, rb)) in enumerate(plan):
                a.halo_peer[k], a.halo_send0[k], a.halo_send1[k], a.halo_recv0[k], a.halo_recv1[k] = int(peer), int(sa), int(sb), int(ra), int(rb)
            a.strip_overlap = 1 if strip_overlap else 0
        elif halo is not None:
            keep.append(halo)
            a.halo = C.cast(halo, C.c_void_p)
            a.strip_overlap = 1 if strip_overlap else 0
    n_samples = (int(spp_range[1]) - int(spp_range[0])) if spp_range is not None else int(spp)
    bvh_restir_worker.ensure_hierarchy_for(float(N) * max(0, n_samples))
    check(lib().mirres_render(ctx.h, bvh_restir_worker.h, C.byref(a), stream_ptr()), "mirres_render")
    return outs, a, keep


class _FusedLoop(torch.autograd.Function):
    """The spp loop of restir_di_with_pt as ONE forward call (mirres_render, batched, several streams) that records a per-sample tape, and ONE backward
    call (mirres_render_bwd) for what the reference differentiates on this path: EvaluateFinalSamples_di + FinalShading of every sample
    (Resampling.py:116-214) — gradients w.r.t. env_map, normal, kd, (roughness, metallic). The indirect sums carry no gradient
    (process_path_tracing_divided_no_grad). Returns the six raw sums of the loop."""
    @staticmethod
    def forward(ctx, env_map, normal_map, diffuse_map, roughness_specular, rctx, worker, mlp_mat, use_scale, scale, occ_map, depth_map, ray_dir_map, pos_map, spp,
                random_offset):
        N = rctx.N
        tape = torch.empty((int(spp) * N, 8), dtype=torch.float32, device=env_map.device)
        sums, a, keep = render_fused(rctx, worker, mlp_mat, use_scale, scale, env_map, occ_map, normal_map, depth_map, diffuse_map, roughness_specular, ray_dir_map,
                                     pos_map, spp, 0, 1, 1.0, 1.0, 1.0, random_offset, spp_range=(0, int(spp)), tape=tape)
        ctx.a, ctx.keep, ctx.rctx, ctx.spp, ctx.env_shape = a, keep, rctx, int(spp), tuple(env_map.shape)
        ctx.mark_non_differentiable(sums[3], sums[4], sums[5])
        return tuple(sums)

    @staticmethod
    def backward(ctx, g_color, g_diff, g_spec, *unused):
        N = ctx.rctx.N
        dev = g_color.device
        z3 = lambda g: _f32(g) if g is not None else torch.zeros((N, 3), dtype=torch.float32, device=dev)
        gc, gd, gs = z3(g_color), z3(g_diff), z3(g_spec)
        need_env, need_n, need_kd, need_rm = ctx.needs_input_grad[0:4]
        g_env = torch.zeros(ctx.env_shape, dtype=torch.float32, device=dev) if need_env else None
        g_n = torch.empty((N, 3), dtype=torch.float32, device=dev) if need_n else None
        g_kd = torch.empty((N, 3), dtype=torch.float32, device=dev) if need_kd else None
        g_rm = torch.empty((N, 2), dtype=torch.float32, device=dev) if need_rm else None
        p = lambda t: t.data_ptr() if t is not None else None
        check(lib().mirres_render_bwd(ctx.rctx.h, C.byref(ctx.a), ctx.spp, gc.data_ptr(), gd.data_ptr(), gs.data_ptr(), p(g_n), p(g_kd), p(g_rm), p(g_env), stream_ptr()),
              "mirres_render_bwd")
        return (g_env, g_n, g_kd, g_rm) + (None,) * 11


def _fused_training():
    import os
    return os.environ.get("MIRRES_TRAIN_FUSED", "1") != "0"


def run_restir_di_with_pt(use_scale, scale_x, scale_y, scale_z, mlp_mat, gb_depth, bvh_restir_worker, make_sampleable_m, generateLightTiles_m, InitialResampling_m,
                          TemporalResampling_m, SpatialResampling_m, EvaluateFinalSamples_m, FinalShading_m, denoising_m, light_data, light_uv, light_inv_pdf,
                          reservoirs, prev_reservoirs, final_samples, neighborOffsets, light_tile_count, light_tile_size, env_map, occ_map, normal_map, depth_map,
                          diffuse_map, roughness_specular, ray_dir_map, pos_map, prev_occ_map, prev_normal_depth, prev_brdf_map, prev_ray_dir, framedim_x,
                          framedim_y, spp, denoise_iter, stepWidth, c_phi_scale=1.0, n_phi_scale=0.1, p_phi_scale=0.1):
    """renderer_restir.py:473-550. Mutates occ_map in place; returns (final_color, denoised_diffuse, denoised_spec, denoised_indirect,
    denoised_indirect_diff, denoised_indirect_spec), each f32[N,3]."""
    from .render_helper import MLPTexture3D
    grad = _needs_grad(env_map, normal_map, diffuse_map, roughness_specular)
    if not grad and (mlp_mat is None or isinstance(mlp_mat, MLPTexture3D)):
        random_offset = np.random.randint(2**20) if _FIXED_RANDOM_OFFSET is None else int(_FIXED_RANDOM_OFFSET)
        outs, _, _ = render_fused(InitialResampling_m.ctx, bvh_restir_worker, mlp_mat, use_scale, (scale_x, scale_y, scale_z), env_map, occ_map, normal_map,
                                  depth_map, diffuse_map, roughness_specular, ray_dir_map, pos_map, spp, denoise_iter, stepWidth, c_phi_scale, n_phi_scale,
                                  p_phi_scale, random_offset, gb_depth=gb_depth)
        return tuple(outs)

    diffuse, normal, roughnessSpecular = diffuse_map, normal_map, roughness_specular
    if _fused_training() and (mlp_mat is None or isinstance(mlp_mat, MLPTexture3D)) and occ_map.is_contiguous():
        # training: the same loop as ONE batched forward + ONE backward call (MIRRES_TRAIN_FUSED=0 selects the reference-shaped sample-by-sample loop below)
        random_offset = np.random.randint(2**20) if _FIXED_RANDOM_OFFSET is None else int(_FIXED_RANDOM_OFFSET)
        (total_color, total_diff_light, total_spec_light, total_color_1, total_diff_light_1, total_spec_light_1) = _FusedLoop.apply(
            env_map, normal_map, diffuse_map, roughness_specular, InitialResampling_m.ctx, bvh_restir_worker, mlp_mat, use_scale, (scale_x, scale_y, scale_z), occ_map,
            depth_map, ray_dir_map, pos_map, spp, random_offset)
        mFrameIndex = spp
    else:
      indices = torch.where(occ_map <= 0.5)
      occ_map[indices[0], :] = 0
      ray_dir_map = safe_l2_normalize(ray_dir_map, dim=-1).contiguous()
      motionVectors = None  # all-zero in the reference (:487)
      color = None
      (total_color, total_color_1, total_diff_light, total_spec_light, total_diff_light_1, total_spec_light_1, total_indirect_light, mFrameIndex) = restir_di_with_pt(
        use_scale, scale_x, scale_y, scale_z, mlp_mat, bvh_restir_worker, spp, framedim_x, framedim_y, make_sampleable_m, generateLightTiles_m, InitialResampling_m,
        TemporalResampling_m, SpatialResampling_m, EvaluateFinalSamples_m, FinalShading_m, light_data, light_uv, light_inv_pdf, reservoirs, prev_reservoirs,
        final_samples, neighborOffsets, light_tile_count, light_tile_size, env_map, occ_map, pos_map, normal, depth_map, diffuse, roughnessSpecular, ray_dir_map,
        prev_occ_map, prev_normal_depth, prev_brdf_map, prev_ray_dir, motionVectors, color)
    total_color = total_color / mFrameIndex
    total_diff_light = total_diff_light / mFrameIndex
    total_spec_light = total_spec_light / mFrameIndex
    total_color_1 = total_color_1 / mFrameIndex
    total_diff_light_1 = total_diff_light_1 / mFrameIndex
    total_spec_light_1 = total_spec_light_1 / mFrameIndex
    combined_color_indirect = total_diff_light_1 + total_spec_light_1
    if gb_depth is None:
        args = (denoising_m, c_phi_scale, n_phi_scale, p_phi_scale, stepWidth, denoise_iter, framedim_x, framedim_y, occ_map)
        denoised_diffuse = EAWDenoise_use_phi(*args, total_diff_light, normal, pos_map)
        denoised_spec = EAWDenoise_use_phi(*args, total_spec_light, normal, pos_map)
        denoised_indirect = EAWDenoise_use_phi_no_di(*args, combined_color_indirect, normal, pos_map)
        denoised_indirect_diff = EAWDenoise_use_phi_no_di(*args, total_diff_light_1, normal, pos_map)
        denoised_indirect_spec = EAWDenoise_use_phi_no_di(*args, total_spec_light_1, normal, pos_map)
    else:   # --use_bi_de (:529-541)
        factor = 2.0
        cat = lambda c: torch.cat((c, normal, gb_depth), dim=-1)
        denoised_diffuse = bilateral_denoiser(framedim_y, framedim_x, cat(total_diff_light), factor)
        denoised_spec = bilateral_denoiser(framedim_y, framedim_x, cat(total_spec_light), factor)
        denoised_indirect = bilateral_denoiser_no_di(framedim_y, framedim_x, cat(combined_color_indirect), factor)
        denoised_indirect_diff = bilateral_denoiser_no_di(framedim_y, framedim_x, cat(total_diff_light_1), factor)
        denoised_indirect_spec = bilateral_denoiser_no_di(framedim_y, framedim_x, cat(total_spec_light_1), factor)
    diffuse = diffuse * (1.0 - roughnessSpecular[..., 1:2])
    final_color = diffuse * denoised_diffuse + denoised_spec + denoised_indirect
    indices = torch.where(occ_map <= 0.1)
    final_color[indices[0], :] = 1.0
    final_color = torch.nan_to_num(final_color, 0.0)
    return final_color, denoised_diffuse, denoised_spec, denoised_indirect, denoised_indirect_diff, denoised_indirect_spec
