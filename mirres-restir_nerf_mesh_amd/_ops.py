"""Thin torch<->C-ABI glue shared by the reference-shaped modules. Every function enqueues on the current torch stream."""
import collections
import ctypes as C
import os

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr


class Module:
    """Stands in for a slangpy module handle (`m`) of the reference: an opaque token that carries the engine context."""

    def __init__(self, name, ctx):
        self.name = name
        self.ctx = ctx

    def __repr__(self):
        return "<mirres module %s>" % self.name

    def process_normal_ao(self, framedim_x, framedim_y, occ_map, normal_map, ray_dir, out_ao):
        """The one kernel the reference launches straight off the denoising module handle (nerf/renderer.py:1153-1158):
        `m.process_normal_ao(framedim_x=..., ..., out_ao=out_ao).launchRaw(blockSize=..., gridSize=...)`.  The launch shape is the engine's own."""
        n = int(framedim_x) * int(framedim_y)
        for name, t, width in (("occ_map", occ_map, 1), ("normal_map", normal_map, 3), ("out_ao", out_ao, 3)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n * width:
                raise _lib.MirresError("process_normal_ao: %s must be a contiguous float32 [%d, %d] tensor" % (name, n, width))
        mod = self

        class _Launch:
            def launchRaw(self, blockSize=None, gridSize=None):
                check(lib().mirres_normal_ao(int(framedim_x), int(framedim_y), occ_map.data_ptr(), normal_map.data_ptr(), out_ao.data_ptr(), stream_ptr()), "mirres_normal_ao")
        return _Launch()


class Context:
    def __init__(self, fx, fy, cfg=None):
        self.fx, self.fy, self.N = int(fx), int(fy), int(fx) * int(fy)
        self.cfg = cfg if cfg is not None else _lib.default_config()
        h = C.c_void_p()
        check(lib().mirres_ctx_create(C.byref(h), self.fx, self.fy, C.byref(self.cfg)), "mirres_ctx_create")
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().mirres_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def stats(self, reset=False):
        out = (C.c_uint64 * 16)()
        check(lib().mirres_ctx_stats(self.h, out, int(reset)), "mirres_ctx_stats")
        v = list(out)
        return dict(rays_any=v[0], rays_closest=v[1], popped=v[2], entered=v[3], leaves=v[4], cl_popped=v[5], cl_entered=v[6], cl_leaves=v[7],
                    any_max_stack=v[8], cl_max_stack=v[9], cl_redo=v[10], any_stack_overflow=v[11], any_dead=v[12],
                    any_wave_iters=v[13], any_wave_leaf_iters=v[14], any_leaf_visits=v[15])      # shadow-ray kernel, counting mode: wave iterations, those that ran the leaf branch, leaf records fetched

    def trace_time(self):
        """(ms_any, launches_any, ms_closest, launches_closest) of the event-timed traversal launches since the last call."""
        a, c = C.c_double(0), C.c_double(0); na, nc = C.c_int(0), C.c_int(0)
        check(lib().mirres_ctx_trace_time(self.h, C.byref(a), C.byref(na), C.byref(c), C.byref(nc)), "mirres_ctx_trace_time")
        return a.value, na.value, c.value, nc.value

    def set_instrument(self, on):
        check(lib().mirres_ctx_set_instrument(self.h, int(on)), "mirres_ctx_set_instrument")

    def reserve(self, samples_per_batch=0):
        """Allocate the K-sample batch pool now (mirres_ctx_reserve) instead of inside the first frame; returns the batch size that fitted."""
        k = lib().mirres_ctx_reserve(self.h, int(samples_per_batch))
        if k < 0:
            check(k, "mirres_ctx_reserve")
        return k


_CTX_CACHE = collections.OrderedDict()      # (device, fx, fy, max_bounce) -> Context, least recently used first


def _ctx_cache_limit():
    """A context lazily owns its batch pool (~690 B x K x N: 113 GB at 1600^2 with the default K = 64 of a 512-spp frame, less for shorter frames; a pool the device cannot hold is halved), so the cache is bounded: MIRRES_CTX_CACHE contexts
    (default 3) per process, least recently used dropped first.  A dropped context is destroyed (pool freed) as soon as nothing else refers to it —
    module handles returned by load_m_for_restir keep theirs alive."""
    try:
        return max(1, int(os.environ.get("MIRRES_CTX_CACHE", "3")))
    except ValueError:
        return 3


def get_ctx(fx, fy, max_bounce=None):
    """The engine context of an fx x fy frame on the current device.  `max_bounce=None` means the configuration default, and both spellings share one
    context (and one pool)."""
    cfg = _lib.default_config()
    if max_bounce is not None:
        cfg.max_bounce = int(max_bounce)
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    key = (dev, int(fx), int(fy), int(cfg.max_bounce))
    ctx = _CTX_CACHE.get(key)
    if ctx is None:
        limit = _ctx_cache_limit()
        while len(_CTX_CACHE) >= limit:
            _CTX_CACHE.popitem(last=False)          # the evicted Context frees its pools in __del__ once unreferenced
        ctx = _CTX_CACHE[key] = Context(fx, fy, cfg)
    else:
        _CTX_CACHE.move_to_end(key)
    return ctx


def _f32(t):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise _lib.MirresError("expected a CUDA float32 tensor, got %s on %s" % (t.dtype, t.device))
    return t if t.is_contiguous() else t.contiguous()


def env_struct(env_tex, width, height, pdf_, cdf_, mpdf_, mcdf_, keep):
    e = _lib.Env()
    for name, t in (("tex", env_tex), ("pdf", pdf_), ("cdf", cdf_), ("mpdf", mpdf_), ("mcdf", mcdf_)):
        if t is not None:
            t = _f32(t); keep.append(t)
            setattr(e, name, t.data_ptr())
    e.Wc, e.Hc = int(width), int(height)
    return e


def gbuf_struct(occ, pos, normal_depth, brdf, ray_dir, keep):
    g = _lib.GBuf()
    for name, t in (("occ", occ), ("pos", pos), ("normal_depth", normal_depth), ("brdf", brdf), ("ray_dir", ray_dir)):
        if t is not None:
            t = _f32(t); keep.append(t)
            setattr(g, name, t.data_ptr())
    return g


def res_struct(reservoirs, keep):
    ld, pdf, M, w = reservoirs
    if M.dtype != torch.int32:
        raise _lib.MirresError("reservoir M must be int32")
    for t in (ld, pdf, M, w):
        if not t.is_contiguous():
            raise _lib.MirresError("reservoir tensors are mutated in place and must be contiguous")
    keep.extend([ld, pdf, M, w])
    r = _lib.Res()
    r.light_data, r.light_pdf, r.M, r.weight = ld.data_ptr(), pdf.data_ptr(), M.data_ptr(), w.data_ptr()
    return r


def path_struct(occ, pos, normal, ray_dir, kd, rm, prd, new_pos, new_ray_d, new_occ, new_normal, keep):
    p = _lib.Path()
    ins = (("occ", occ), ("pos", pos), ("normal", normal), ("ray_dir", ray_dir), ("kd", kd), ("rough_metal", rm))
    outs = (("prd", prd), ("new_pos", new_pos), ("new_ray_d", new_ray_d), ("new_occ", new_occ), ("new_normal", new_normal))
    for name, t in ins:
        t = _f32(t); keep.append(t); setattr(p, name, t.data_ptr())
    for name, t in outs:
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise _lib.MirresError("%s is written in place and must be a contiguous float32 tensor" % name)
        keep.append(t); setattr(p, name, t.data_ptr())
    return p
