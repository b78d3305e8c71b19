"""ctypes binding of libmirres.so (include/mirres.h). Fails loudly when the HIP library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIRRES_LIB: another build of the same library (A/B measurements of kernel variants); it must export the full ABI like the in-tree one
LIB_PATH = os.environ.get("MIRRES_LIB") or os.path.join(_HERE, "libmirres.so")
_lib = None

vp = C.c_void_p
i32 = C.c_int32
u32 = C.c_uint32
f32 = C.c_float


class MirresError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("light_tile_count", C.c_int), ("light_tile_size", C.c_int), ("screen_tile_size", C.c_int),
                ("initial_light_samples", C.c_int), ("initial_brdf_samples", C.c_int), ("max_history", C.c_int),
                ("neighbor_offset_count", C.c_int), ("neighbor_count", C.c_int), ("gather_radius", C.c_float),
                ("max_bounce", C.c_int), ("vis_near", C.c_float)]


class Env(C.Structure):
    _fields_ = [("tex", vp), ("Wc", C.c_int), ("Hc", C.c_int), ("pdf", vp), ("cdf", vp), ("mpdf", vp), ("mcdf", vp)]


class GBuf(C.Structure):
    _fields_ = [("occ", vp), ("pos", vp), ("normal_depth", vp), ("brdf", vp), ("ray_dir", vp)]


class Res(C.Structure):
    _fields_ = [("light_data", vp), ("light_pdf", vp), ("M", vp), ("weight", vp)]


class Path(C.Structure):
    _fields_ = [(n, vp) for n in ("occ", "pos", "normal", "ray_dir", "kd", "rough_metal", "prd", "new_pos", "new_ray_d", "new_occ", "new_normal")]


class MatNet(C.Structure):
    _fields_ = [("grid_f16", vp), ("w0", vp), ("w1", vp), ("w2", vp),
                ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("out_min", C.c_float * 6), ("out_max", C.c_float * 6)]


class RenderArgs(C.Structure):
    _fields_ = [("spp", C.c_int), ("random_offset", C.c_uint32), ("use_scale", C.c_int), ("scale", C.c_float * 3),
                ("env_map", vp), ("Wc", C.c_int), ("Hc", C.c_int),
                ("occ", vp), ("normal", vp), ("depth", vp), ("kd", vp), ("rough_metal", vp), ("ray_dir", vp), ("pos", vp),
                ("mat", C.POINTER(MatNet)), ("const_kd", C.c_float * 3), ("const_rm", C.c_float * 2),
                ("denoise_iter", C.c_int), ("step_width", C.c_int), ("c_phi", C.c_float), ("n_phi", C.c_float), ("p_phi", C.c_float),
                ("outs", vp * 6), ("tape", vp), ("gb_depth", vp), ("spp_begin", C.c_int), ("spp_end", C.c_int),
                ("strip_full_fy", C.c_int), ("strip_y_off", C.c_int), ("own_y0", C.c_int), ("own_y1", C.c_int), ("halo", vp), ("halo_user", vp), ("strip_overlap", C.c_int),
                ("halo_comm", vp), ("halo_n", C.c_int), ("halo_peer", C.c_int * 2), ("halo_send0", C.c_int * 2), ("halo_send1", C.c_int * 2), ("halo_recv0", C.c_int * 2),
                ("halo_recv1", C.c_int * 2), ("halo_time_stride", C.c_int)]


HALO_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int, vp)   # int halo(void* user, float* records, int sample, void* stream)


# every symbol include/mirres.h declares: name -> (restype, argtypes)
PCFG, PENV, PG, PRES, PPATH, PMAT, PARGS = (C.POINTER(t) for t in (Config, Env, GBuf, Res, Path, MatNet, RenderArgs))
SIGNATURES = {
    "mirres_default_config": (None, [PCFG]),
    "mirres_version": (C.c_char_p, []),
    "mirres_last_error": (C.c_char_p, []),
    "mirres_bvh_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "mirres_bvh_destroy": (None, [vp]),
    "mirres_bvh_build": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]),
    "mirres_bvh_build_level": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp]),
    "mirres_bvh_upgrade": (C.c_int, [vp, vp, vp, vp, vp, vp]),
    "mirres_bvh_private_level": (C.c_int, [vp]),
    "mirres_bvh_trace": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "mirres_ctx_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, PCFG]),
    "mirres_comm_unique_id": (C.c_int, [C.c_char_p, vp]),
    "mirres_comm_create": (C.c_int, [C.POINTER(vp), C.c_char_p, vp, C.c_int, C.c_int]),
    "mirres_comm_destroy": (None, [vp]),
    "mirres_ctx_halo_time": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mirres_ctx_destroy": (None, [vp]),
    "mirres_neighbor_offsets": (C.c_int, [vp, vp, vp]),
    "mirres_ctx_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
    "mirres_ctx_set_instrument": (C.c_int, [vp, C.c_int]),
    "mirres_ctx_trace_time": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mirres_env_make_sampleable": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "mirres_light_tiles": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, u32, vp, vp, vp, vp]),
    "mirres_restir_initial": (C.c_int, [vp, vp, PENV, PG, PRES, vp, vp, u32, vp]),
    "mirres_restir_temporal": (C.c_int, [vp, PENV, PG, PG, PRES, PRES, vp, u32, vp]),
    "mirres_restir_spatial": (C.c_int, [vp, vp, PENV, PG, PRES, PRES, vp, u32, vp]),
    "mirres_restir_final_vis": (C.c_int, [vp, vp, vp, PRES, vp, vp]),
    "mirres_restir_eval_final": (C.c_int, [vp, PENV, PRES, vp, vp, vp, vp, vp]),
    "mirres_restir_eval_final_bwd": (C.c_int, [vp, PENV, PRES, vp, vp, vp, vp]),
    "mirres_final_shading": (C.c_int, [vp, PENV] + [vp] * 11 + [vp]),
    "mirres_final_shading_bwd": (C.c_int, [vp] + [vp] * 15 + [vp]),
    "mirres_pt_new_dir": (C.c_int, [vp, vp, PPATH, u32, u32, vp]),
    "mirres_pt_bounce": (C.c_int, [vp, vp, PENV, PPATH, u32, u32, vp, vp, vp, vp]),
    "mirres_eaw": (C.c_int, [C.c_int, C.c_int, C.c_int, f32, f32, f32, vp, vp, vp, vp, vp, vp]),
    "mirres_normal_ao": (C.c_int, [C.c_int, C.c_int, vp, vp, vp, vp]),
    "mirres_prepare_shading_normal": (C.c_int, [C.c_longlong, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "mirres_prepare_shading_normal_bwd": (C.c_int, [C.c_longlong, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mirres_raster_raycast": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp]),
    "mirres_interpolate": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp, vp]),
    "mirres_rasterize": (C.c_int, [vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, vp, vp, vp]),
    "mirres_interpolate_bwd": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp]),
    "mirres_texture2d": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp]),
    "mirres_texture2d_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]),
    "mirres_antialias": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "mirres_antialias_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp]),
    "mirres_dump_render": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "mirres_bilateral": (C.c_int, [C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp, vp]),
    "mirres_bilateral_bwd": (C.c_int, [C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp, vp]),
    "mirres_eaw_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mirres_eaw_bwd_gather": (C.c_int, [C.c_int, C.c_int, C.c_int, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mirres_matnet_grid_entries": (C.c_int, []),
    "mirres_matnet_pack_grid": (C.c_int, [vp, vp, C.c_int64, vp]),
    "mirres_matnet_fwd": (C.c_int, [PMAT, vp, C.c_int, vp, vp, vp]),
    "mirres_matnet_mlp": (C.c_int, [PMAT, vp, C.c_int, vp, vp]),
    "mirres_matnet_scatter": (C.c_int, [PMAT, vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_float), vp]),
    "mirres_matnet_bwd": (C.c_int, [PMAT, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "mirres_ctx_reserve": (C.c_int, [vp, C.c_int]),
    "mirres_render": (C.c_int, [vp, vp, PARGS, vp]),
    "mirres_render_bwd": (C.c_int, [vp, PARGS, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mirres_render_finish": (C.c_int, [vp, PARGS, C.POINTER(vp), vp]),
    "mirres_selfcheck_arith": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), vp]),
    "mirres_fmath_eval": (C.c_int, [C.c_int, vp, vp, vp, C.c_longlong, vp]),
    "mirres_fmath_checksum": (C.c_int, [C.c_int, C.c_uint, C.c_uint64, C.POINTER(C.c_uint64), vp]),
}


def lib():
    """Returns the loaded library; raises MirresError (never falls back) if it is missing or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MirresError("libmirres.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or mirres-restir_nerf_mesh_amd/csrc/build.py — there is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing:
            raise MirresError("libmirres.so lacks symbols declared in include/mirres.h: %s" % ", ".join(missing))
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise MirresError("%s failed (%d): %s" % (what, rc, lib().mirres_last_error().decode()))


def default_config():
    c = Config()
    lib().mirres_default_config(C.byref(c))
    return c


def ptr(t):
    """Device pointer of a torch tensor (must be contiguous) or None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise MirresError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
