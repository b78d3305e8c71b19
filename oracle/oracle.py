"""ORACLE — TEST INFRASTRUCTURE ONLY. ctypes binding of oracle/liborc.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: see oracle/orc_math.hpp.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)
u64p = C.POINTER(C.c_ulonglong)


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))] + [os.path.join(_HERE, "..", "include", "mirres_fmath.h")]
    if force or not os.path.exists(so) or not os.path.exists(os.path.join(_HERE, "libfmathcheck.so")) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def build_ref(force=False):
    """oracle/_ref: the reference's own bilateral-denoiser kernels compiled by hipcc from /root/reference (oracle/Makefile `ref`). Only possible where the
    reference tree exists (the build container); the GPU box uses the prebuilt oracle/_ref/libref_denoise.so that travelled with the snapshot."""
    ref = os.environ.get("MIRRES_REF", "/root/reference")
    if not os.path.isdir(ref):
        return None
    out = os.path.join(_HERE, "_ref", "libref_denoise.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(os.path.join(_HERE, "ref_denoise_driver.hip")):
        subprocess.run(["make", "-C", _HERE, "ref", "REF=" + ref] + (["-B"] if force else []), check=True, capture_output=True, text=True)   # the tree the isdir test looked at
    return out


def ref_denoise_lib():
    """ctypes handle of oracle/_ref/libref_denoise.so (GPU code: ref_bilateral_fwd / ref_bilateral_bwd take device pointers), or None when it was not built."""
    p = os.path.join(_HERE, "_ref", "libref_denoise.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    vp = C.c_void_p
    L.ref_bilateral_fwd.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]; L.ref_bilateral_fwd.restype = C.c_int
    L.ref_bilateral_bwd.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]; L.ref_bilateral_bwd.restype = C.c_int
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_next1d.restype = C.c_float
        _LIB.orc_f16_to_f32.restype = C.c_float
        _LIB.orc_seed.restype = C.c_uint32
        _LIB.orc_expand_bits.restype = C.c_uint32
        _LIB.orc_morton3d.restype = C.c_uint32
        _LIB.orc_morton3d.argtypes = [C.c_float] * 3
        _LIB.orc_f32_to_f16.argtypes = [C.c_float]
        _LIB.orc_f32_to_f16.restype = C.c_uint16
    return _LIB


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Frame(C.Structure):
    _fields_ = [("fx", C.c_int), ("fy", C.c_int),
                ("occ", f32p), ("pos", f32p), ("normal_depth", f32p), ("brdf", f32p), ("ray_dir", f32p),
                ("info", i32p), ("aabb", f32p), ("vert", f32p), ("tri", i32p),
                ("env_tex", f32p), ("env_w", C.c_int), ("env_h", C.c_int),
                ("pdf", f32p), ("cdf", f32p), ("mpdf", f32p), ("mcdf", f32p), ("max_bounce", C.c_int),
                ("neighbor_count", C.c_int), ("initial_light_samples", C.c_int), ("max_history", C.c_int)]


class Res(C.Structure):
    _fields_ = [("light_data", f32p), ("light_pdf", f32p), ("M", i32p), ("weight", f32p)]


class Path(C.Structure):
    _fields_ = [(n, f32p) for n in ("occ", "pos", "normal", "ray_dir", "kd", "rs", "prd", "new_pos", "new_ray_d", "new_occ", "new_normal")]


class MatNet(C.Structure):
    _fields_ = [("params_f16", u16p), ("w0", f32p), ("w1", f32p), ("w2", f32p),
                ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("mn", C.c_float * 6), ("mx", C.c_float * 6)]


class RenderArgs(C.Structure):
    _fields_ = [("fx", C.c_int), ("fy", C.c_int), ("spp", C.c_int), ("random_offset", C.c_uint32), ("max_bounce", C.c_int),
                ("use_scale", C.c_int), ("scale", C.c_float * 3),
                ("info", i32p), ("aabb", f32p), ("vert", f32p), ("tri", i32p),
                ("env_map", f32p), ("env_w", C.c_int), ("env_h", C.c_int),
                ("occ", f32p), ("normal", f32p), ("depth", f32p), ("kd", f32p), ("rs", f32p), ("ray_dir", f32p), ("pos", f32p),
                ("mat", C.POINTER(MatNet)), ("const_kd", C.c_float * 3), ("const_rs", C.c_float * 2),
                ("denoise_iter", C.c_int), ("step_width", C.c_int), ("c_phi", C.c_float), ("n_phi", C.c_float), ("p_phi", C.c_float),
                ("outs", f32p * 6), ("counters", u64p), ("ray_count", u64p), ("avg_direct", f32p)]


# ------------------------------------------------------------------ scalar helpers
def seed(px, py, n):
    return int(lib().orc_seed(C.c_uint32(px), C.c_uint32(py), C.c_uint32(n)))


def next1d(state):
    s = C.c_uint32(state)
    v = lib().orc_next1d(C.byref(s))
    return float(v), int(s.value)


def expand_bits(v):
    return int(lib().orc_expand_bits(C.c_uint32(v)))


def morton3d(x, y, z):
    return int(lib().orc_morton3d(x, y, z))


def oct_encode(n):
    n = _c(n, np.float32); out = np.zeros(2, np.float32)
    lib().orc_oct_encode(_p(n, f32p), _p(out, f32p)); return out


def oct_decode(f):
    f = _c(f, np.float32); out = np.zeros(3, np.float32)
    lib().orc_oct_decode(_p(f, f32p), _p(out, f32p)); return out


# ------------------------------------------------------------------ BVH
def bvh_build(vert, tri):
    vert = _c(vert, np.float32); tri = _c(tri, np.int32)
    T = tri.shape[0]
    info = np.zeros((2 * T - 1, 3), np.int32); aabb = np.zeros((2 * T - 1, 6), np.float32)
    srt = np.zeros((T, 2), np.int32); mh = C.c_int(0)
    rc = lib().orc_bvh_build(_p(vert, f32p), vert.shape[0], _p(tri, i32p), T, _p(info, i32p), _p(aabb, f32p), _p(srt, i32p), C.byref(mh))
    assert rc == 0
    return info, aabb, srt, mh.value


def make_rays(o, d, tmin=0.0, tmax=1e7):
    o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
    n = o.shape[0]
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = o; r[:, 3] = tmin; r[:, 4:7] = d; r[:, 7] = tmax
    return r


def trace(info, aabb, vert, tri, rays, want_normal=True, counters=False):
    rays = _c(rays, np.float32); n = rays.shape[0]
    hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); pos = np.zeros((n, 3), np.float32)
    nrm = np.zeros((n, 3), np.float32); prim = np.zeros(n, np.int32)
    cnt = np.zeros((n, 4), np.uint32) if counters else None
    lib().orc_trace(_p(info, i32p), _p(aabb, f32p), _p(_c(vert, np.float32), f32p), _p(_c(tri, np.int32), i32p), _p(rays, f32p), n,
                    int(want_normal), _p(hit, i32p), _p(t, f32p), _p(pos, f32p), _p(nrm, f32p), _p(prim, i32p), _p(cnt, u32p))
    out = dict(hit=hit, t=t, pos=pos, normal=nrm, prim=prim)
    if counters:
        out["counters"] = cnt
    return out


def trace_stack_depth(info, aabb, vert, tri, rays, want_normal=False):
    """Most entries the reference's 64-entry stack holds for each ray (helperDi.slang:136, :197-274)."""
    rays = _c(rays, np.float32); n = rays.shape[0]
    out = np.zeros(n, np.uint32)
    lib().orc_trace_stack_depth(_p(info, i32p), _p(aabb, f32p), _p(_c(vert, np.float32), f32p), _p(_c(tri, np.int32), i32p), _p(rays, f32p), n, int(want_normal), _p(out, u32p))
    return out


def tree_depth(info):
    """Depth (edges on the longest root-to-leaf path) of the LBVH in the reference's `info` layout (left, right, prim per node; root 0)."""
    info = np.asarray(info).reshape(-1, 3)
    depth = np.zeros(len(info), np.int32)
    front = np.array([0]); d = 0
    while len(front):
        depth[front] = d
        kids = info[front, :2].reshape(-1)
        front = kids[kids != 0]
        d += 1
    return int(depth.max())


# ------------------------------------------------------------------ environment
def flip_env(env_map):
    """renderer_restir.py:305-311: vertical flip + flatten to [Hc*Wc,3]."""
    return np.ascontiguousarray(env_map[::-1].reshape(-1, 3), dtype=np.float32)


def make_sampleable(tex_flat, W, H):
    pdf = np.zeros(W * H, np.float32); cdf = np.zeros((W + 1) * H, np.float32); mpdf = np.zeros(H, np.float32); mcdf = np.zeros(H + 1, np.float32)
    lib().orc_make_sampleable(_p(_c(tex_flat, np.float32), f32p), W, H, _p(pdf, f32p), _p(cdf, f32p), _p(mpdf, f32p), _p(mcdf, f32p))
    return pdf, cdf, mpdf, mcdf


def env_weights(tex_flat, W, H):
    out = np.zeros(W * H, np.float32)
    lib().orc_env_weights(_p(_c(tex_flat, np.float32), f32p), W, H, _p(out, f32p))
    return out


def distribution2d(pdf, cdf, W, H):
    pdf = _c(pdf, np.float32).copy(); cdf = _c(cdf, np.float32).copy()
    lib().orc_distribution2d(W, H, _p(pdf, f32p), _p(cdf, f32p))
    return pdf, cdf


def neighbor_offsets(count=8192):
    out = np.zeros(2 * count, np.float32)
    lib().orc_neighbor_offsets(count, _p(out, f32p))
    return out.reshape(-1, 2)


def env_le(tex_flat, W, H, dirs):
    dirs = _c(dirs, np.float32); out = np.zeros_like(dirs)
    lib().orc_env_le(_p(_c(tex_flat, np.float32), f32p), W, H, _p(dirs, f32p), dirs.shape[0], _p(out, f32p))
    return out


class Keep:
    """Holds numpy arrays alive next to the ctypes struct that points into them."""
    def __init__(self):
        self.refs = []

    def __call__(self, a, dt=np.float32):
        a = _c(a, dt); self.refs.append(a); return a


def make_frame(keep, fx, fy, occ, pos, normal_depth, brdf, ray_dir, bvh, vert, tri, env_tex, env_w, env_h, tables, max_bounce=2,
               neighbor_count=0, initial_light_samples=0, max_history=0):
    f = Frame()
    f.fx, f.fy = fx, fy
    f.occ = _p(keep(occ), f32p); f.pos = _p(keep(pos), f32p); f.normal_depth = _p(keep(normal_depth), f32p)
    f.brdf = _p(keep(brdf), f32p); f.ray_dir = _p(keep(ray_dir), f32p)
    f.info = _p(keep(bvh[0], np.int32), i32p); f.aabb = _p(keep(bvh[1]), f32p)
    f.vert = _p(keep(vert), f32p); f.tri = _p(keep(tri, np.int32), i32p)
    f.env_tex = _p(keep(env_tex), f32p); f.env_w, f.env_h = env_w, env_h
    f.pdf, f.cdf, f.mpdf, f.mcdf = (_p(keep(t), f32p) for t in tables)
    f.max_bounce = max_bounce
    f.neighbor_count, f.initial_light_samples, f.max_history = neighbor_count, initial_light_samples, max_history   # 0: the reference's constants
    return f


def new_reservoirs(N):
    return [np.zeros((N, 3), np.float32), np.zeros(N, np.float32), np.zeros(N, np.int32), np.zeros(N, np.float32)]


def res_struct(r):
    s = Res(); s.light_data = _p(r[0], f32p); s.light_pdf = _p(r[1], f32p); s.M = _p(r[2], i32p); s.weight = _p(r[3], f32p); return s


def light_tiles(frame, frameIndex, tile_count=128, tile_size=1024):
    n = tile_count * tile_size
    ld = np.zeros((n, 3), np.float32); uv = np.zeros((n, 2), np.int32); pdf = np.zeros(n, np.float32)
    lib().orc_light_tiles(C.byref(frame), C.c_uint32(frameIndex), tile_count, tile_size, _p(ld, f32p), _p(uv, i32p), _p(pdf, f32p))
    return ld, uv, pdf


def initial(frame, res, tile_data, tile_pdf, frameIndex, counters=None):
    rs = res_struct(res)
    lib().orc_initial(C.byref(frame), C.byref(rs), _p(tile_data, f32p), _p(tile_pdf, f32p), C.c_uint32(frameIndex), _p(counters, u64p))


def temporal(frame, res, prev, p_occ, p_nd, p_brdf, p_rd, frameIndex, motion=None):
    rs, ps = res_struct(res), res_struct(prev)
    lib().orc_temporal(C.byref(frame), C.byref(rs), C.byref(ps), _p(p_occ, f32p), _p(p_nd, f32p), _p(p_brdf, f32p), _p(p_rd, f32p),
                       _p(motion, f32p), C.c_uint32(frameIndex))


def spatial(frame, res, prev, noff, frameIndex, counters=None):
    rs, ps = res_struct(res), res_struct(prev)
    lib().orc_spatial(C.byref(frame), C.byref(rs), C.byref(ps), _p(noff, f32p), C.c_uint32(frameIndex), _p(counters, u64p))


def final_vis(frame, res, counters=None):
    N = frame.fx * frame.fy; vis = np.ones(N, np.float32); rs = res_struct(res)
    lib().orc_final_vis(C.byref(frame), C.byref(rs), _p(vis, f32p), _p(counters, u64p))
    return vis


def eval_final(frame, res, vis):
    N = frame.fx * frame.fy
    fdir = np.zeros((N, 3), np.float32); fdist = np.zeros(N, np.float32); fLi = np.zeros((N, 3), np.float32); rs = res_struct(res)
    lib().orc_eval_final(C.byref(frame), C.byref(rs), _p(vis, f32p), _p(fdir, f32p), _p(fdist, f32p), _p(fLi, f32p))
    return fdir, fdist, fLi


def final_shading(frame, normal, kd, rs_map, fdir, fdist, fLi):
    N = frame.fx * frame.fy
    c = np.zeros((N, 3), np.float32); d = np.zeros((N, 3), np.float32); s = np.zeros((N, 3), np.float32)
    lib().orc_final_shading(C.byref(frame), _p(_c(normal, np.float32), f32p), _p(_c(kd, np.float32), f32p), _p(_c(rs_map, np.float32), f32p),
                            _p(fdir, f32p), _p(fdist, f32p), _p(fLi, f32p), _p(c, f32p), _p(d, f32p), _p(s, f32p))
    return c, d, s


def path_struct(occ, pos, normal, ray_dir, kd, rs, prd, new_pos, new_ray_d, new_occ, new_normal):
    p = Path()
    for n, a in zip(("occ", "pos", "normal", "ray_dir", "kd", "rs", "prd", "new_pos", "new_ray_d", "new_occ", "new_normal"),
                    (occ, pos, normal, ray_dir, kd, rs, prd, new_pos, new_ray_d, new_occ, new_normal)):
        assert a.dtype == np.float32 and a.flags.c_contiguous
        setattr(p, n, _p(a, f32p))
    return p


def new_dir(frame, path, frameIndex, bounce_count=0, counters=None):
    lib().orc_new_dir(C.byref(frame), C.byref(path), C.c_uint32(frameIndex), C.c_uint32(bounce_count), _p(counters, u64p))


def bounce(frame, path, frameIndex, bounce_count, counters=None):
    N = frame.fx * frame.fy
    c = np.zeros((N, 3), np.float32); d = np.zeros((N, 3), np.float32); s = np.zeros((N, 3), np.float32)
    lib().orc_bounce(C.byref(frame), C.byref(path), C.c_uint32(frameIndex), C.c_uint32(bounce_count), _p(c, f32p), _p(d, f32p), _p(s, f32p), _p(counters, u64p))
    return c, d, s


def eaw(fx, fy, step, c_phi, n_phi, p_phi, occ, color, normal, pos):
    out = np.zeros((fx * fy, 3), np.float32)
    lib().orc_eaw(fx, fy, int(step), C.c_float(c_phi), C.c_float(n_phi), C.c_float(p_phi), _p(_c(occ, np.float32), f32p), _p(_c(color, np.float32), f32p),
                  _p(_c(normal, np.float32), f32p), _p(_c(pos, np.float32), f32p), _p(out, f32p))
    return out


def normal_ao(fx, fy, occ, normal):
    """process_normal_ao (EAWDenoise.slang:591-651)."""
    out = np.zeros((fx * fy, 3), np.float32)
    lib().orc_normal_ao(int(fx), int(fy), _p(_c(occ, np.float32), f32p), _p(_c(normal, np.float32), f32p), _p(out, f32p))
    return out


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True):
    """nerf/renderutils: bsdf_prepare_shading_normal (ops.py:84-121) == c_src/normal.cu forward, restated in numpy float32 (test infrastructure).
    safeNormalize = v / |v| (0 for the zero vector, vec3f.h:87-91); bend threshold 0.1 (normal.cu:12)."""
    f = np.float32
    def nz(v):
        l = np.sqrt(((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]).astype(f))[..., None]
        return np.where(l > 0, v / np.where(l > 0, l, f(1)), f(0)).astype(f)
    dot = lambda a, b: ((a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]).astype(f)[..., None]
    pos, view_pos, p, sn, st, gn = (np.asarray(a, f) for a in (pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm))
    nrm, tng, view = nz(sn), nz(st), nz((view_pos - pos).astype(f))
    bit = nz(np.cross(tng, nrm).astype(f))
    sgn = f(-1.0 if opengl else 1.0)
    sh = nz((tng * p[..., 0:1] + bit * (sgn * p[..., 1:2]) + nrm * np.maximum(p[..., 2:3], f(0))).astype(f))
    flip = (dot(view, gn) < 0) if two_sided_shading else np.zeros(dot(view, gn).shape, bool)
    sh2, gn2 = np.where(flip, -sh, sh), np.where(flip, -gn, gn)
    t = np.clip(dot(view, sh2) / f(0.1), f(0), f(1)).astype(f)
    return (gn2 * (f(1) - t) + sh2 * t).astype(f)


def bilateral(fx, fy, sigma, col, nrm, zdz, grad4=None):
    """grad4 None: forward -> f32[N,4] (sum w col, max(sum w, 1e-4)); else backward -> col_grad f32[N,3]."""
    n = fx * fy
    nrm = _c(nrm, np.float32); zdz = _c(zdz, np.float32)
    if grad4 is None:
        out = np.zeros((n, 4), np.float32)
        lib().orc_bilateral(fx, fy, C.c_float(sigma), _p(_c(col, np.float32), f32p), _p(nrm, f32p), _p(zdz, f32p), None, 0, _p(out, f32p))
    else:
        out = np.zeros((n, 3), np.float32)
        lib().orc_bilateral(fx, fy, C.c_float(sigma), None, _p(nrm, f32p), _p(zdz, f32p), _p(_c(grad4, np.float32), f32p), 1, _p(out, f32p))
    return out


# ------------------------------------------------------------------ BRDF library probes (tests/test_brdf_invariants.py)
def sh_lobes(kd, rough, metal, ray_dir, normal):
    """Lobe probabilities, GGX alpha and F0 colour as FinalShading.slang:58-78 derives them: (pD, pS, alpha, specular[n,3])."""
    kd = _c(kd, np.float32); n = kd.shape[0]
    pD = np.zeros(n, np.float32); pS = np.zeros(n, np.float32); al = np.zeros(n, np.float32); sp = np.zeros((n, 3), np.float32)
    lib().orc_sh_lobes(n, _p(kd, f32p), _p(_c(rough, np.float32), f32p), _p(_c(metal, np.float32), f32p), _p(_c(ray_dir, np.float32), f32p), _p(_c(normal, np.float32), f32p),
                       _p(pD, f32p), _p(pS, f32p), _p(al, f32p), _p(sp, f32p))
    return pD, pS, al, sp


def sh_eval(pD, pS, alpha, spec_albedo, diff_albedo, wo, wi):
    """utils/brdfDi.slang in the local frame: dict(f = FalcorBRDF_eval [n,3], pdf = FalcorBRDF_evalPdf, spec_f = SpecularReflection_eval, spec_pdf, diff_light = Diffuse_light)."""
    wo = _c(wo, np.float32); n = wo.shape[0]
    b = lambda a, w: _c(np.broadcast_to(np.asarray(a, np.float32), (n,) if w == 1 else (n, w)), np.float32)
    f = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); sf = np.zeros((n, 3), np.float32); spdf = np.zeros(n, np.float32); dl = np.zeros(n, np.float32)
    lib().orc_sh_eval(n, _p(b(pD, 1), f32p), _p(b(pS, 1), f32p), _p(b(alpha, 1), f32p), _p(b(spec_albedo, 3), f32p), _p(b(diff_albedo, 3), f32p), _p(wo, f32p), _p(b(wi, 3), f32p),
                      _p(f, f32p), _p(pdf, f32p), _p(sf, f32p), _p(spdf, f32p), _p(dl, f32p))
    return dict(f=f, pdf=pdf, spec_f=sf, spec_pdf=spdf, diff_light=dl)


def sh_sample(sg, pD, pS, alpha, spec_albedo, diff_albedo, wo, with_weight=True):
    """FalcorBRDF_sample (with_weight) / FalcorBRDF_sample_no_weight for every generator state in `sg` (uint32): dict(wi, pdf, specular_bounce, weight, valid, sg_out, u_select)."""
    sg = _c(sg, np.uint32); n = sg.shape[0]
    wi = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); sb = np.zeros(n, np.uint32); wt = np.zeros((n, 3), np.float32)
    valid = np.zeros(n, np.int32); so = np.zeros(n, np.uint32); us = np.zeros(n, np.float32)
    lib().orc_sh_sample(n, _p(sg, u32p), C.c_float(pD), C.c_float(pS), C.c_float(alpha), _p(_c(spec_albedo, np.float32), f32p), _p(_c(diff_albedo, np.float32), f32p),
                        _p(_c(wo, np.float32), f32p), int(with_weight), _p(wi, f32p), _p(pdf, f32p), _p(sb, u32p), _p(wt, f32p), _p(valid, i32p), _p(so, u32p), _p(us, f32p))
    return dict(wi=wi, pdf=pdf, specular_bounce=sb, weight=wt, valid=valid.astype(bool), sg_out=so, u_select=us)


def rt_eval(L, V, N, alpha, wd, ws):
    """utils/brdf.slang: (evalBRDF, evalPdfBRDF(specularOnly=false)) per row, world space."""
    L = _c(L, np.float32); n = L.shape[0]
    b = lambda a, w: _c(np.broadcast_to(np.asarray(a, np.float32), (n,) if w == 1 else (n, w)), np.float32)
    f = np.zeros(n, np.float32); pdf = np.zeros(n, np.float32)
    lib().orc_rt_eval(n, _p(L, f32p), _p(b(V, 3), f32p), _p(b(N, 3), f32p), _p(b(alpha, 1), f32p), _p(b(wd, 1), f32p), _p(b(ws, 1), f32p), _p(f, f32p), _p(pdf, f32p))
    return f, pdf


def rt_sample(xi, V, N, alpha, wd, ws):
    """utils/brdf.slang sampleBRDF(specularOnly=false): (dir [n,3], valid)."""
    xi = _c(xi, np.float32); n = xi.shape[0]
    d = np.zeros((n, 3), np.float32); valid = np.zeros(n, np.int32)
    lib().orc_rt_sample(n, _p(xi, f32p), _p(_c(V, np.float32), f32p), _p(_c(N, np.float32), f32p), C.c_float(alpha), C.c_float(wd), C.c_float(ws), _p(d, f32p), _p(valid, i32p))
    return d, valid.astype(bool)


def sh_frame(n):
    """create_frame (helperDi.slang:18-30): tangent x, y of the shading frame around normal n."""
    x = np.zeros(3, np.float32); y = np.zeros(3, np.float32)
    lib().orc_sh_frame(_p(_c(n, np.float32), f32p), _p(x, f32p), _p(y, f32p))
    return x, y


# ------------------------------------------------------------------ material field
def hashgrid_layout():
    off = np.zeros(17, np.uint32); res = np.zeros(16, np.uint32); sc = np.zeros(16, np.float32)
    total = lib().orc_hashgrid_layout(_p(off, u32p), _p(res, u32p), _p(sc, f32p))
    return total, off, res, sc


def to_f16_bits(a):
    a = _c(a, np.float32).ravel(); out = np.zeros(a.size, np.uint16)
    lib().orc_f32_to_f16_array(_p(a, f32p), _p(out, u16p), C.c_longlong(a.size))
    return out


def matnet_struct(keep, params_f32, w0, w1, w2, aabb_min, aabb_max, mn, mx):
    m = MatNet()
    m.params_f16 = _p(keep(to_f16_bits(params_f32), np.uint16), u16p)
    m.w0 = _p(keep(w0), f32p); m.w1 = _p(keep(w1), f32p); m.w2 = _p(keep(w2), f32p)
    m.aabb_min[:] = list(map(float, aabb_min)); m.aabb_max[:] = list(map(float, aabb_max))
    m.mn[:] = list(map(float, mn)); m.mx[:] = list(map(float, mx))
    return m


def hashgrid_encode(mat, x01):
    x01 = _c(x01, np.float32); n = x01.shape[0]; out = np.zeros((n, 32), np.uint16)
    lib().orc_hashgrid_encode(C.byref(mat), _p(x01, f32p), n, _p(out, u16p))
    return out


def matnet(mat, pos):
    pos = _c(pos, np.float32); n = pos.shape[0]; out = np.zeros((n, 6), np.float32)
    lib().orc_matnet(C.byref(mat), _p(pos, f32p), n, _p(out, f32p))
    return out


# ------------------------------------------------------------------ whole frame
def set_render_constants(neighbor_count=0, initial_light_samples=0, max_history=0):
    """ReSTIR constants of the following render() calls (0 = the reference's compile-time value); reset with no arguments."""
    lib().orc_set_render_constants(int(neighbor_count), int(initial_light_samples), int(max_history))


def set_dead_ray_override(mode=-1):
    """Test hook (orc_kernels.hpp g_dead_ray_override): 0 / 1 = spatial shadow rays aimed at the sample of a zero-weight reservoir are not traced and report
    free / occluded; 2 = control: every spatial shadow ray reports occluded; -1 = the reference's behaviour. Returns the number of rays the hook answered since the
    previous call."""
    f = lib().orc_set_dead_ray_override; f.restype = C.c_longlong
    return int(f(int(mode)))


def render(fx, fy, spp, random_offset, bvh, vert, tri, env_map, occ, normal, depth, kd, rs, ray_dir, pos, mat=None, max_bounce=2,
           use_scale=False, scale=(1, 1, 1), denoise_iter=2, step_width=2, c_phi=2.0, n_phi=0.1, p_phi=0.001, want_avg=False,
           const_kd=(0.6, 0.6, 0.6), const_rs=(0.5, 0.0)):
    keep = Keep(); N = fx * fy
    a = RenderArgs()
    a.fx, a.fy, a.spp, a.random_offset, a.max_bounce = fx, fy, spp, random_offset, max_bounce
    a.use_scale = int(use_scale); a.scale[:] = list(map(float, scale))
    a.info = _p(keep(bvh[0], np.int32), i32p); a.aabb = _p(keep(bvh[1]), f32p); a.vert = _p(keep(vert), f32p); a.tri = _p(keep(tri, np.int32), i32p)
    env_map = keep(env_map); a.env_map = _p(env_map, f32p); a.env_h, a.env_w = env_map.shape[0], env_map.shape[1]
    occ = keep(np.array(occ, np.float32).reshape(-1).copy()); a.occ = _p(occ, f32p)
    a.normal = _p(keep(normal), f32p); a.depth = _p(keep(depth), f32p); a.kd = _p(keep(kd), f32p); a.rs = _p(keep(rs), f32p)
    a.ray_dir = _p(keep(ray_dir), f32p); a.pos = _p(keep(pos), f32p)
    a.mat = C.pointer(mat) if mat is not None else None
    a.const_kd[:] = list(map(float, const_kd)); a.const_rs[:] = list(map(float, const_rs))
    a.denoise_iter, a.step_width, a.c_phi, a.n_phi, a.p_phi = denoise_iter, step_width, c_phi, n_phi, p_phi
    outs = [np.zeros((N, 3), np.float32) for _ in range(6)]
    for i in range(6):
        a.outs[i] = _p(outs[i], f32p)
    cnt = np.zeros(4, np.uint64); a.counters = _p(cnt, u64p); a.ray_count = None
    avg = np.zeros((N, 3), np.float32) if want_avg else None
    a.avg_direct = _p(avg, f32p)
    rc = lib().orc_render(C.byref(a)); assert rc == 0
    return dict(final_color=outs[0], diffuse=outs[1], spec=outs[2], indirect=outs[3], indirect_diff=outs[4], indirect_spec=outs[5],
                counters=cnt, occ=occ, avg_direct=avg)


def finish(fx, fy, spp, occ, normal, pos, kd, rs, sums, denoise_iter=2, step_width=2, c_phi=2.0, n_phi=0.1, p_phi=0.001):
    """Second half of run_restir_di_with_pt (renderer_restir.py:507-549) on six raw sums [6,N,3] -> the six outputs [6,N,3]. `occ` already thresholded."""
    N = fx * fy
    s_ = [_c(np.array(a, np.float32, copy=True), np.float32) for a in sums]
    o_ = [np.zeros((N, 3), np.float32) for _ in range(6)]
    sp = (f32p * 6)(*[_p(a, f32p) for a in s_]); op = (f32p * 6)(*[_p(a, f32p) for a in o_])
    lib().orc_finish(fx, fy, int(spp), _p(_c(occ, np.float32), f32p), _p(_c(normal, np.float32), f32p), _p(_c(pos, np.float32), f32p), _p(_c(kd, np.float32), f32p),
                     _p(_c(rs, np.float32), f32p), int(denoise_iter), int(step_width), C.c_float(c_phi), C.c_float(n_phi), C.c_float(p_phi), sp, op)
    return o_


# ------------------------------------------------------------------ nerf/render_dump.py (BASELINE configs[0])
def envir_map_dirs(envmap_h, envmap_w):
    """generate_envir_map_dir (nerf/render_helper.py:8-26, is_jittor=False): lat-long light set, z up; (area weights [H*W], directions [H*W,3])."""
    lat = np.pi / envmap_h; lng = 2 * np.pi / envmap_w
    phi, theta = np.meshgrid(np.linspace(np.pi / 2 - 0.5 * lat, -np.pi / 2 + 0.5 * lat, envmap_h, dtype=np.float32),
                             np.linspace(np.pi - 0.5 * lng, -np.pi + 0.5 * lng, envmap_w, dtype=np.float32), indexing="ij")
    sin_phi = np.sin(np.float32(np.pi / 2) - phi)
    w = (4 * np.float32(np.pi) * sin_phi / np.sum(sin_phi, dtype=np.float32)).astype(np.float32).reshape(-1)
    d = np.stack([np.cos(theta) * np.cos(phi), np.sin(theta) * np.cos(phi), np.sin(phi)], -1).reshape(-1, 3).astype(np.float32)
    return w, d


def occluded_front(info, aabb, vert, tri, rays):
    """A conventional occlusion query (hit in front of the origin) over the oracle's hierarchy: what render_dump.py expects of its `intersector`."""
    rays = _c(rays, np.float32); n = rays.shape[0]; hit = np.zeros(n, np.int32)
    lib().orc_occluded_front(_p(info, i32p), _p(aabb, f32p), _p(_c(vert, np.float32), f32p), _p(_c(tri, np.int32), i32p), _p(rays, f32p), n, _p(hit, i32p))
    return hit


def dump_light_rgbs(env_map, H, W, dirs):
    env = _c(np.asarray(env_map, np.float32).reshape(H, W, 3), np.float32); dirs = _c(dirs, np.float32); L = dirs.shape[0]
    out = np.zeros((L, 3), np.float32)
    lib().orc_dump_light_rgbs(_p(env, f32p), H, W, _p(dirs, f32p), L, _p(out, f32p))
    return out


def ggx_specular(normal, pts2c, pts2l, rough, fresnel):
    normal = _c(normal, np.float32); n = normal.shape[0]; pts2l = _c(pts2l, np.float32); L = pts2l.shape[0]
    out = np.zeros((n, L, 3), np.float32)
    lib().orc_ggx_specular(n, L, _p(normal, f32p), _p(_c(pts2c, np.float32), f32p), _p(pts2l, f32p), _p(_c(rough, np.float32), f32p), _p(_c(fresnel, np.float32), f32p), _p(out, f32p))
    return out


def dump_render(bvh, vert, tri, pos, normal, albedo, rough, fresnel, rays_d, env_map, env_h, env_w, light_w, light_dirs, equal_areas=False, clamp_rgb=True):
    """dump_render (render_dump.py:84-133) with this oracle's front-only occlusion query as the intersector. Returns (rgb, diff, spec), each [n,3]."""
    info, aabb = bvh
    pos = _c(pos, np.float32); n = pos.shape[0]; light_dirs = _c(light_dirs, np.float32); L = light_dirs.shape[0]
    lrgb = dump_light_rgbs(env_map, env_h, env_w, light_dirs)
    o = [np.zeros((n, 3), np.float32) for _ in range(3)]
    lib().orc_dump_render(_p(info, i32p), _p(aabb, f32p), _p(_c(vert, np.float32), f32p), _p(_c(tri, np.int32), i32p), n, L, _p(pos, f32p), _p(_c(normal, np.float32), f32p),
                          _p(_c(albedo, np.float32), f32p), _p(_c(rough, np.float32), f32p), _p(_c(fresnel, np.float32), f32p), _p(_c(rays_d, np.float32), f32p),
                          _p(light_dirs, f32p), _p(_c(light_w, np.float32), f32p), _p(lrgb, f32p), int(equal_areas), int(clamp_rgb), _p(o[0], f32p), _p(o[1], f32p), _p(o[2], f32p))
    return o


def num_threads():
    return int(lib().orc_num_threads())
