// normal.hip — prepare_shading_normal of nerf/renderutils (ops.py:100-163, c_src/normal.cu): the shading normal render_stage1 feeds to the path
// (nerf/renderer.py:1013). Elementwise: tangent-space perturbation of the interpolated normal, two-sided flip against the geometric normal,
// bending of back-facing normals towards the viewer (threshold 0.1). Forward and a hand-derived reverse mode; fp32, IEEE div/sqrt.
#include "engine.hpp"
#include "device_math.hpp"

namespace mr {

#define MR_BLOCK 256
#define MR_NRM_THRESHOLD 0.1f

MR_DEV v3 safe_normalize(v3 v) { const float l = sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z); return l > 0.f ? v / l : V3(0.f); }
// cotangent of safe_normalize: (g - n (n . g)) / |v|
MR_DEV v3 safe_normalize_bwd(v3 v, v3 g) {
    const float l = sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
    if (!(l > 0.f)) return V3(0.f);
    const v3 n = v / l;
    return (g - n * dot(n, g)) / l;
}
MR_DEV v3 perturb(v3 p, v3 nrm, v3 tng, float sgn) {
    const v3 bit = safe_normalize(cross(tng, nrm));
    return safe_normalize(tng * p.x + bit * (sgn * p.y) + nrm * fmaxf(p.z, 0.f));
}
MR_DEV v3 bend(v3 view, v3 smooth, v3 geom) {
    const float t = clampf(dot(view, smooth) / MR_NRM_THRESHOLD, 0.f, 1.f);
    return geom * (1.0f - t) + smooth * t;
}

struct NrmIn { const float *pos, *view_pos, *perturbed, *smooth_nrm, *smooth_tng, *geom_nrm; };

__global__ void __launch_bounds__(MR_BLOCK) k_shading_normal(size_t n, NrmIn I, int two_sided, int opengl, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v3 nrm = safe_normalize(ld3(I.smooth_nrm, i)), tng = safe_normalize(ld3(I.smooth_tng, i));
    const v3 view = safe_normalize(ld3(I.view_pos, i) - ld3(I.pos, i));
    const v3 geom = ld3(I.geom_nrm, i);
    const v3 sh = perturb(ld3(I.perturbed, i), nrm, tng, opengl ? -1.f : 1.f);
    const bool flip = two_sided && dot(view, geom) < 0.f;
    st3(out, i, flip ? bend(view, -sh, -geom) : bend(view, sh, geom));
}

__global__ void __launch_bounds__(MR_BLOCK) k_shading_normal_bwd(size_t n, NrmIn I, int two_sided, int opengl, const float* __restrict__ dout, float* __restrict__ g_pos,
                                                                 float* __restrict__ g_view_pos, float* __restrict__ g_perturbed, float* __restrict__ g_smooth_nrm,
                                                                 float* __restrict__ g_smooth_tng, float* __restrict__ g_geom) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v3 rn = ld3(I.smooth_nrm, i), rt = ld3(I.smooth_tng, i), rv = ld3(I.view_pos, i) - ld3(I.pos, i);
    const v3 nrm = safe_normalize(rn), tng = safe_normalize(rt), view = safe_normalize(rv);
    const v3 geom0 = ld3(I.geom_nrm, i), p = ld3(I.perturbed, i);
    const float sgn = opengl ? -1.f : 1.f;
    // forward intermediates
    const v3 rb = cross(tng, nrm), bit = safe_normalize(rb);
    const v3 rs = tng * p.x + bit * (sgn * p.y) + nrm * fmaxf(p.z, 0.f);
    const v3 sh0 = safe_normalize(rs);
    const bool flip = two_sided && dot(view, geom0) < 0.f;
    const v3 sh = flip ? -sh0 : sh0, geom = flip ? -geom0 : geom0;
    const float dp = dot(view, sh), t = clampf(dp / MR_NRM_THRESHOLD, 0.f, 1.f);
    // reverse: out = geom (1 - t) + sh t
    const v3 go = ld3(dout, i);
    v3 d_geom = go * (1.0f - t), d_sh = go * t, d_view = V3(0.f);
    const float d_t = dot(go, sh - geom);
    const float d_dp = (dp < 0.f || dp > MR_NRM_THRESHOLD) ? 0.f : d_t / MR_NRM_THRESHOLD;     // clamp passes the gradient only inside (0, threshold)
    d_view = d_view + sh * d_dp; d_sh = d_sh + view * d_dp;
    if (flip) { d_sh = -d_sh; d_geom = -d_geom; }
    // sh0 = normalize(rs), rs = tng p.x + bit sgn p.y + nrm max(p.z, 0)
    const v3 d_rs = safe_normalize_bwd(rs, d_sh);
    v3 d_tng = d_rs * p.x, d_bit = d_rs * (sgn * p.y), d_nrm = p.z > 0.f ? d_rs * p.z : V3(0.f);
    const v3 d_p = V3(dot(d_rs, tng), sgn * dot(d_rs, bit), p.z > 0.f ? dot(d_rs, nrm) : 0.f);
    // bit = normalize(rb), rb = tng x nrm:  d_tng += nrm x d_rb,  d_nrm += d_rb x tng
    const v3 d_rb = safe_normalize_bwd(rb, d_bit);
    d_tng = d_tng + cross(nrm, d_rb); d_nrm = d_nrm + cross(d_rb, tng);
    const v3 d_rv = safe_normalize_bwd(rv, d_view);
    if (g_pos) st3(g_pos, i, -d_rv);
    if (g_view_pos) st3(g_view_pos, i, d_rv);
    if (g_perturbed) st3(g_perturbed, i, d_p);
    if (g_smooth_nrm) st3(g_smooth_nrm, i, safe_normalize_bwd(rn, d_nrm));
    if (g_smooth_tng) st3(g_smooth_tng, i, safe_normalize_bwd(rt, d_tng));
    if (g_geom) st3(g_geom, i, d_geom);
}

}  // namespace mr

using namespace mr;

extern "C" {

int mirres_prepare_shading_normal(long long n, const float* pos, const float* view_pos, const float* perturbed_nrm, const float* smooth_nrm, const float* smooth_tng,
                                  const float* geom_nrm, int two_sided_shading, int opengl, float* out, void* stream) {
    if (n < 0 || !pos || !view_pos || !perturbed_nrm || !smooth_nrm || !smooth_tng || !geom_nrm || !out) { set_error("mirres_prepare_shading_normal: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    NrmIn I = {pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm};
    k_shading_normal<<<grid_for((size_t)n, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>((size_t)n, I, two_sided_shading, opengl, out);
    MR_LAUNCH_CHECK("prepare_shading_normal");
    return MIRRES_OK;
}

int mirres_prepare_shading_normal_bwd(long long n, const float* pos, const float* view_pos, const float* perturbed_nrm, const float* smooth_nrm, const float* smooth_tng,
                                      const float* geom_nrm, int two_sided_shading, int opengl, const float* dout, float* g_pos, float* g_view_pos,
                                      float* g_perturbed_nrm, float* g_smooth_nrm, float* g_smooth_tng, float* g_geom_nrm, void* stream) {
    if (n < 0 || !pos || !view_pos || !perturbed_nrm || !smooth_nrm || !smooth_tng || !geom_nrm || !dout) { set_error("mirres_prepare_shading_normal_bwd: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    NrmIn I = {pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm};
    k_shading_normal_bwd<<<grid_for((size_t)n, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>((size_t)n, I, two_sided_shading, opengl, dout, g_pos, g_view_pos, g_perturbed_nrm,
                                                                                            g_smooth_nrm, g_smooth_tng, g_geom_nrm);
    MR_LAUNCH_CHECK("prepare_shading_normal_bwd");
    return MIRRES_OK;
}

}  // extern "C"
