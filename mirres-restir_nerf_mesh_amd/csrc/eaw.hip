// eaw.hip — edge-avoiding a-trous wavelet denoiser (process_EAWDenoise / _no_di, EAWDenoise.slang:50-302) + hand-derived adjoint.
// 25 taps of a B3 spline (weights k[ix]*k[iy]/256, k = 1,4,6,4,1 — the literal table at :114-143), edge weights
// exp(-|dc|^2/c_phi) * exp(-|dn|^2/n_phi) * exp(-|dp|^2/p_phi); background pixels are copied. HBM-bound: 25 x 36 B of
// gathers per pixel that hit L2 (neighbouring threads read neighbouring pixels).
#include "engine.hpp"
#include "device_math.hpp"

namespace mr {

#define MR_BLOCK 256

MR_DEV float b3(int i) { return i == 2 ? 6.f : ((i == 1 || i == 3) ? 4.f : 1.f); }

__global__ void __launch_bounds__(MR_BLOCK) k_eaw(int fx, int fy, int step, float c_phi, float n_phi, float p_phi, const float* __restrict__ occ,
                                                  const float* __restrict__ color, const float* __restrict__ normal, const float* __restrict__ pos,
                                                  float* __restrict__ out) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= fx * fy) return;
    if (occ[pi] < 0.1f) { st3(out, pi, ld3(color, pi)); return; }
    const int x = pi % fx, y = pi / fx;
    const v3 nval = ld3(normal, pi), pval = ld3(pos, pi), cval = ld3(color, pi);
    v3 sum = V3(0.f); float cum_w = 0.0f;
#pragma unroll 5
    for (int i = 0; i < 25; i++) {
        const int ox = (i % 5) - 2, oy = (i / 5) - 2;
        const int ux = x + (int)((float)ox * step), uy = y + (int)((float)oy * step);
        if (!(ux >= 0 && uy >= 0 && ux < fx && uy < fy)) continue;
        const size_t qi = (size_t)uy * fx + ux;
        const float kw = b3(i % 5) * b3(i / 5) / 256.0f;
        const v3 ctmp = ld3(color, qi);
        v3 t = cval - ctmp;
        float dist2 = dot(t, t);
        const float c_w = fminf(expf(-(dist2) / c_phi), 1.0f);
        t = nval - ld3(normal, qi);
        dist2 = fmaxf(dot(t, t), 0.0f);
        const float n_w = fminf(expf(-(dist2) / n_phi), 1.0f);
        t = pval - ld3(pos, qi);
        dist2 = fmaxf(dot(t, t), 0.0f);
        const float p_w = fminf(expf(-(dist2) / p_phi), 1.0f);
        const float weight = c_w * n_w * p_w;
        sum = sum + ctmp * weight * kw;
        cum_w += weight * kw;
    }
    st3(out, pi, sum / cum_w);
}

MR_DEV void atomic_add3(float* p, size_t i, v3 v) { atomicAdd(&p[3 * i], v.x); atomicAdd(&p[3 * i + 1], v.y); atomicAdd(&p[3 * i + 2], v.z); }

// Adjoint of k_eaw (what Slang autodiff produces for process_EAWDenoise.bwd, Denoising.py:38-44):
//   out = S / W,  S = sum_i c_i w_i k_i,  W = sum_i w_i k_i,  w_i = exp(-|c0-c_i|^2/phi_c) exp(-|n0-n_i|^2/phi_n) exp(-|p0-p_i|^2/phi_p)
__global__ void __launch_bounds__(MR_BLOCK) k_eaw_bwd(int fx, int fy, int step, float c_phi, float n_phi, float p_phi, const float* __restrict__ occ,
                                                      const float* __restrict__ color, const float* __restrict__ normal, const float* __restrict__ pos,
                                                      const float* __restrict__ gout, float* __restrict__ gc, float* __restrict__ gn, float* __restrict__ gp) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= fx * fy) return;
    const v3 g = ld3(gout, pi);
    if (occ[pi] < 0.1f) { atomic_add3(gc, pi, g); return; }
    const int x = pi % fx, y = pi / fx;
    const v3 nval = ld3(normal, pi), pval = ld3(pos, pi), cval = ld3(color, pi);
    // pass 1: recompute S and W
    v3 S = V3(0.f); float W = 0.f;
    for (int i = 0; i < 25; i++) {
        const int ux = x + (int)((float)((i % 5) - 2) * step), uy = y + (int)((float)((i / 5) - 2) * step);
        if (!(ux >= 0 && uy >= 0 && ux < fx && uy < fy)) continue;
        const size_t qi = (size_t)uy * fx + ux;
        const float kw = b3(i % 5) * b3(i / 5) / 256.0f;
        const v3 ctmp = ld3(color, qi);
        v3 t = cval - ctmp; const float cw = fminf(expf(-dot(t, t) / c_phi), 1.0f);
        t = nval - ld3(normal, qi); const float nw = fminf(expf(-fmaxf(dot(t, t), 0.f) / n_phi), 1.0f);
        t = pval - ld3(pos, qi); const float pw = fminf(expf(-fmaxf(dot(t, t), 0.f) / p_phi), 1.0f);
        const float w = cw * nw * pw;
        S = S + ctmp * w * kw; W += w * kw;
    }
    const v3 outv = S / W;
    const v3 gS = g / W;
    const float gW = -dot(g, outv) / W;
    v3 g_c0 = V3(0.f), g_n0 = V3(0.f), g_p0 = V3(0.f);
    for (int i = 0; i < 25; i++) {
        const int ux = x + (int)((float)((i % 5) - 2) * step), uy = y + (int)((float)((i / 5) - 2) * step);
        if (!(ux >= 0 && uy >= 0 && ux < fx && uy < fy)) continue;
        const size_t qi = (size_t)uy * fx + ux;
        const float kw = b3(i % 5) * b3(i / 5) / 256.0f;
        const v3 ctmp = ld3(color, qi);
        const v3 tc = cval - ctmp; const float cw = fminf(expf(-dot(tc, tc) / c_phi), 1.0f);
        const v3 tn = nval - ld3(normal, qi); const float nw = fminf(expf(-fmaxf(dot(tn, tn), 0.f) / n_phi), 1.0f);
        const v3 tp = pval - ld3(pos, qi); const float pw = fminf(expf(-fmaxf(dot(tp, tp), 0.f) / p_phi), 1.0f);
        const float w = cw * nw * pw;
        v3 g_ci = gS * (w * kw);                       // through the tap value in S
        const float g_w = (dot(gS, ctmp) + gW) * kw;   // through w_i in S and W
        // w = cw*nw*pw, each factor exp(-d2/phi): d w / d d2_x = -w/phi_x
        const v3 gtc = tc * (2.f * g_w * (-w / c_phi));
        const v3 gtn = tn * (2.f * g_w * (-w / n_phi));
        const v3 gtp = tp * (2.f * g_w * (-w / p_phi));
        g_c0 = g_c0 + gtc; g_ci = g_ci - gtc;
        g_n0 = g_n0 + gtn; g_p0 = g_p0 + gtp;
        atomic_add3(gc, qi, g_ci);
        if (gn) atomic_add3(gn, qi, -gtn);
        if (gp) atomic_add3(gp, qi, -gtp);
    }
    atomic_add3(gc, pi, g_c0);
    if (gn) atomic_add3(gn, pi, g_n0);
    if (gp) atomic_add3(gp, pi, g_p0);
}

}  // namespace mr

using namespace mr;

extern "C" {

int mirres_eaw(int fx, int fy, int step_width, float c_phi, float n_phi, float p_phi, const float* occ, const float* color, const float* normal,
               const float* pos, float* out, void* stream) {
    if (fx <= 0 || fy <= 0 || !occ || !color || !normal || !pos || !out || color == out) { set_error("mirres_eaw: bad argument"); return MIRRES_E_ARG; }
    k_eaw<<<grid_for((size_t)fx * fy, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(fx, fy, step_width, c_phi, n_phi, p_phi, occ, color, normal, pos, out);
    MR_LAUNCH_CHECK("eaw");
    return MIRRES_OK;
}

int mirres_eaw_bwd(int fx, int fy, int step_width, float c_phi, float n_phi, float p_phi, const float* occ, const float* color, const float* normal,
                   const float* pos, const float* grad_out, float* g_color, float* g_normal, float* g_pos, void* stream) {
    if (fx <= 0 || fy <= 0 || !occ || !color || !normal || !pos || !grad_out || !g_color) { set_error("mirres_eaw_bwd: bad argument"); return MIRRES_E_ARG; }
    k_eaw_bwd<<<grid_for((size_t)fx * fy, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(fx, fy, step_width, c_phi, n_phi, p_phi, occ, color, normal, pos, grad_out,
                                                                                         g_color, g_normal, g_pos);
    MR_LAUNCH_CHECK("eaw_bwd");
    return MIRRES_OK;
}

}  // extern "C"
