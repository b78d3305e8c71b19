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
        const float c_w = fminf(mrf_exp(-(dist2) / c_phi), 1.0f);
        t = nval - ld3(normal, qi);
        dist2 = fmaxf(dot(t, t), 0.0f);
        const float n_w = fminf(mrf_exp(-(dist2) / n_phi), 1.0f);
        t = pval - ld3(pos, qi);
        dist2 = fmaxf(dot(t, t), 0.0f);
        const float p_w = fminf(mrf_exp(-(dist2) / p_phi), 1.0f);
        const float weight = c_w * n_w * p_w;
        sum = sum + ctmp * weight * kw;
        cum_w += weight * kw;
    }
    st3(out, pi, sum / cum_w);
}

// The frame's five a-trous runs (diffuse, specular, indirect, indirect diffuse, indirect specular: run_restir_di_with_pt :521-533) share the
// guides, hence the tap geometry and the normal / position weights; only the colour weight differs per buffer. One pass per iteration: per tap
// 24 B of guides + 5 x 12 B of colours instead of 5 x 36 B, 7 exponentials instead of 15 — each buffer's result is the expression k_eaw
// evaluates, bit for bit (weight = c_w * n_w * p_w in that order).
struct Eaw5 { const float* in[5]; float* out[5]; };
__global__ void __launch_bounds__(MR_BLOCK) k_eaw5(int fx, int fy, int step, float c_phi, float n_phi, float p_phi, const float* __restrict__ occ,
                                                   const float* __restrict__ normal, const float* __restrict__ pos, Eaw5 B) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= fx * fy) return;
    if (occ[pi] < 0.1f) {
#pragma unroll
        for (int b = 0; b < 5; b++) st3(B.out[b], pi, ld3(B.in[b], pi));
        return;
    }
    const int x = pi % fx, y = pi / fx;
    const v3 nval = ld3(normal, pi), pval = ld3(pos, pi);
    v3 cval[5], sum[5]; float cum_w[5];
#pragma unroll
    for (int b = 0; b < 5; b++) { cval[b] = ld3(B.in[b], pi); sum[b] = V3(0.f); cum_w[b] = 0.0f; }
#pragma unroll 5
    for (int i = 0; i < 25; i++) {
        const int ox = (i % 5) - 2, oy = (i / 5) - 2;
        const int ux = x + (int)((float)ox * step), uy = y + (int)((float)oy * step);
        if (!(ux >= 0 && uy >= 0 && ux < fx && uy < fy)) continue;
        const size_t qi = (size_t)uy * fx + ux;
        const float kw = b3(i % 5) * b3(i / 5) / 256.0f;
        v3 t = nval - ld3(normal, qi);
        float dist2 = fmaxf(dot(t, t), 0.0f);
        const float n_w = fminf(mrf_exp(-(dist2) / n_phi), 1.0f);
        t = pval - ld3(pos, qi);
        dist2 = fmaxf(dot(t, t), 0.0f);
        const float p_w = fminf(mrf_exp(-(dist2) / p_phi), 1.0f);
#pragma unroll
        for (int b = 0; b < 5; b++) {
            const v3 ctmp = ld3(B.in[b], qi);
            const v3 tc = cval[b] - ctmp;
            const float c_w = fminf(mrf_exp(-(dot(tc, tc)) / c_phi), 1.0f);
            const float weight = c_w * n_w * p_w;
            sum[b] = sum[b] + ctmp * weight * kw;
            cum_w[b] += weight * kw;
        }
    }
#pragma unroll
    for (int b = 0; b < 5; b++) st3(B.out[b], pi, sum[b] / cum_w[b]);
}
int launch_eaw5(int fx, int fy, int step, float c_phi, float n_phi, float p_phi, const float* occ, const float* const in[5], const float* normal, const float* pos,
                float* const out[5], hipStream_t s) {
    Eaw5 B; for (int b = 0; b < 5; b++) { B.in[b] = in[b]; B.out[b] = out[b]; }
    k_eaw5<<<grid_for((size_t)fx * fy, MR_BLOCK), MR_BLOCK, 0, s>>>(fx, fy, step, c_phi, n_phi, p_phi, occ, normal, pos, B);
    MR_LAUNCH_CHECK("eaw5");
    return 0;
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
        v3 t = cval - ctmp; const float cw = fminf(mrf_exp(-dot(t, t) / c_phi), 1.0f);
        t = nval - ld3(normal, qi); const float nw = fminf(mrf_exp(-fmaxf(dot(t, t), 0.f) / n_phi), 1.0f);
        t = pval - ld3(pos, qi); const float pw = fminf(mrf_exp(-fmaxf(dot(t, t), 0.f) / p_phi), 1.0f);
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
        const v3 tc = cval - ctmp; const float cw = fminf(mrf_exp(-dot(tc, tc) / c_phi), 1.0f);
        const v3 tn = nval - ld3(normal, qi); const float nw = fminf(mrf_exp(-fmaxf(dot(tn, tn), 0.f) / n_phi), 1.0f);
        const v3 tp = pval - ld3(pos, qi); const float pw = fminf(mrf_exp(-fmaxf(dot(tp, tp), 0.f) / p_phi), 1.0f);
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


// The same adjoint as a GATHER (no atomics, deterministic): the tap set is symmetric (offset o and -o, equal spline weight) and the edge weight of
// a pair (p, r) is a symmetric function of the two pixels, so one loop over p's 25 neighbours r yields both what p contributes as the CENTRE of its
// own filter (needs gS_p, gW_p) and what it receives as a TAP of r's filter (needs gS_r, gW_r). Pass 1 stores (gS, gW) per pixel (16 B); pass 2
// gathers. k_eaw_bwd above issues up to 225 atomics per pixel: 0.67 ms for a 640 k-pixel buffer against ~0.1 ms for the two passes.
__global__ void __launch_bounds__(MR_BLOCK) k_eaw_bwd_sums(int fx, int fy, int step, float c_phi, float n_phi, float p_phi, const float* __restrict__ occ,
                                                           const float* __restrict__ color, const float* __restrict__ normal, const float* __restrict__ pos,
                                                           const float* __restrict__ gout, float4* __restrict__ sums) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= fx * fy) return;
    if (occ[pi] < 0.1f) { sums[pi] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const int x = pi % fx, y = pi / fx;
    const v3 g = ld3(gout, pi), nval = ld3(normal, pi), pval = ld3(pos, pi), cval = ld3(color, pi);
    v3 S = V3(0.f); float W = 0.f;
    for (int i = 0; i < 25; i++) {
        const int ux = x + (int)((float)((i % 5) - 2) * step), uy = y + (int)((float)((i / 5) - 2) * step);
        if (!(ux >= 0 && uy >= 0 && ux < fx && uy < fy)) continue;
        const size_t qi = (size_t)uy * fx + ux;
        const float kw = b3(i % 5) * b3(i / 5) / 256.0f;
        const v3 ctmp = ld3(color, qi);
        v3 t = cval - ctmp; const float cw = fminf(mrf_exp(-dot(t, t) / c_phi), 1.0f);
        t = nval - ld3(normal, qi); const float nw = fminf(mrf_exp(-fmaxf(dot(t, t), 0.f) / n_phi), 1.0f);
        t = pval - ld3(pos, qi); const float pw = fminf(mrf_exp(-fmaxf(dot(t, t), 0.f) / p_phi), 1.0f);
        const float w = cw * nw * pw;
        S = S + ctmp * w * kw; W += w * kw;
    }
    const v3 outv = S / W;
    const v3 gS = g / W;
    sums[pi] = make_float4(gS.x, gS.y, gS.z, -dot(g, outv) / W);
}
__global__ void __launch_bounds__(MR_BLOCK) k_eaw_bwd_gather(int fx, int fy, int step, float c_phi, float n_phi, float p_phi, const float* __restrict__ occ,
                                                             const float* __restrict__ color, const float* __restrict__ normal, const float* __restrict__ pos,
                                                             const float* __restrict__ gout, const float4* __restrict__ sums, float* __restrict__ gc,
                                                             float* __restrict__ gn, float* __restrict__ gp) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= fx * fy) return;
    const int x = pi % fx, y = pi / fx;
    const bool fg = !(occ[pi] < 0.1f);
    const v3 nval = ld3(normal, pi), pval = ld3(pos, pi), cval = ld3(color, pi);
    const float4 sp = sums[pi];
    const v3 gS_p = V3(sp.x, sp.y, sp.z); const float gW_p = sp.w;
    v3 a_c = fg ? V3(0.f) : ld3(gout, pi), a_n = V3(0.f), a_p = V3(0.f);      // a background pixel is copied by the forward
    for (int i = 0; i < 25; i++) {
        const int ux = x + (int)((float)((i % 5) - 2) * step), uy = y + (int)((float)((i / 5) - 2) * step);
        if (!(ux >= 0 && uy >= 0 && ux < fx && uy < fy)) continue;
        const size_t ri = (size_t)uy * fx + ux;
        const bool rfg = !(occ[ri] < 0.1f);
        if (!fg && !rfg) continue;
        const float kw = b3(i % 5) * b3(i / 5) / 256.0f;
        const v3 cr = ld3(color, ri);
        const v3 tc = cval - cr; const float cw = fminf(mrf_exp(-dot(tc, tc) / c_phi), 1.0f);
        const v3 tn = nval - ld3(normal, ri); const float nw = fminf(mrf_exp(-fmaxf(dot(tn, tn), 0.f) / n_phi), 1.0f);
        const v3 tp = pval - ld3(pos, ri); const float pw = fminf(mrf_exp(-fmaxf(dot(tp, tp), 0.f) / p_phi), 1.0f);
        const float w = cw * nw * pw;
        if (fg) {      // p as the centre, r as its tap: the terms k_eaw_bwd keeps in g_c0 / g_n0 / g_p0
            const float g_w = (dot(gS_p, cr) + gW_p) * kw;
            a_c = a_c + tc * (2.f * g_w * (-w / c_phi)); a_n = a_n + tn * (2.f * g_w * (-w / n_phi)); a_p = a_p + tp * (2.f * g_w * (-w / p_phi));
        }
        if (rfg) {     // r as the centre, p as its tap (offset -o, same spline weight, same edge weight): what k_eaw_bwd scatters to the tap
            const float4 sr = sums[ri];
            const v3 gS_r = V3(sr.x, sr.y, sr.z);
            const float g_w = (dot(gS_r, cval) + sr.w) * kw;
            // differences seen from r: (c_r - c_p) = -tc, ...; the tap receives  gS_r w kw - gtc_r  with gtc_r = (-tc) 2 g_w (-w / c_phi)
            a_c = a_c + gS_r * (w * kw) + tc * (2.f * g_w * (-w / c_phi));
            a_n = a_n + tn * (2.f * g_w * (-w / n_phi));
            a_p = a_p + tp * (2.f * g_w * (-w / p_phi));
        }
    }
    st3(gc, pi, ld3(gc, pi) + a_c);
    if (gn) st3(gn, pi, ld3(gn, pi) + a_n);
    if (gp) st3(gp, pi, ld3(gp, pi) + a_p);
}

// ---------------------------------------------------------------- bilateral denoiser (nerf/renderutils: ops.py:173-211, c_src/denoising.cu:14-130)
// The alternative denoiser of run_restir_di_with_pt (--use_bi_de, renderer_restir.py:529-541). Per pixel, over a (2r+1)^2 window with
// r = 2 ceil(2.5 sigma) + 1 (sigma = 4 -> 43 x 43 taps): w = exp(-d^2 / 2 sigma^2) * clamp(n_t . n_c, 1e-4, 1)^128 * exp(-|z_t - z_c| / max(dz_c d, 1e-4));
// out = (sum w col, max(sum w, 1e-4)). The backward is the transposed gather (the depth term's denominator uses the TAP's dz, denoising.cu:113).
// Inputs are read from one packed 32-byte record per pixel {col.xyz n.x | n.yz z dz} built by k_bilateral_pack (the normal is normalised
// there: safe_normalize, ops.py:168-169), so a tap is two 16-byte loads that neighbouring pixels share through L1.
#define MR_BIL_EPS 0.0001f
__global__ void __launch_bounds__(MR_BLOCK) k_bilateral_pack(size_t n, const float* __restrict__ col, const float* __restrict__ nrm, const float* __restrict__ zdz,
                                                             float4* __restrict__ rec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 nn = ld3(nrm, i);
    const float len = sqrtf(fmaxf(dot(nn, nn), 1e-20f));
    nn = V3(nn.x / len, nn.y / len, nn.z / len);
    float4 a, b;
    a.x = col ? col[3 * i] : 0.f; a.y = col ? col[3 * i + 1] : 0.f; a.z = col ? col[3 * i + 2] : 0.f; a.w = nn.x;
    b.x = nn.y; b.y = nn.z; b.z = zdz[2 * i]; b.w = zdz[2 * i + 1];
    rec[2 * i] = a; rec[2 * i + 1] = b;
}
// MODE 0: forward, out4 = (sum w col, max(sum w, 1e-4)); MODE 1: forward divided, out3 = sum w col / max(sum w, 1e-4) (the fused frame loop);
// MODE 2: backward, out3 = sum w' grad_out4[tap].xyz
template <int MODE>
__global__ void __launch_bounds__(MR_BLOCK) k_bilateral(int fx, int fy, float sigma, const float4* __restrict__ rec, const float* __restrict__ grad_out4,
                                                        float* __restrict__ out) {
    // 16 x 16 pixel tiles: the 58 x 58 tap window of a workgroup stays in L1 / L2
    const int tiles_x = (fx + 15) >> 4;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 16 + (int)(threadIdx.x & 15), y = ty * 16 + (int)(threadIdx.x >> 4);
    if (x >= fx || y >= fy) return;
    const size_t pi = (size_t)y * fx + x;
    const float4 ca = rec[2 * pi], cb = rec[2 * pi + 1];
    const v3 c_nrm = V3(ca.w, cb.x, cb.y);
    const float c_z = cb.z, c_dz = cb.w;
    const float variance = sigma * sigma;
    const int rad = 2 * (int)ceilf(sigma * 2.5f) + 1;
    float accum_w = 0.f; v3 acc = V3(0.f);
    for (int dy = -rad; dy <= rad; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= fy) continue;
        for (int dx = -rad; dx <= rad; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= fx) continue;
            const size_t qi = (size_t)yy * fx + xx;
            const float4 ta = rec[2 * qi], tb = rec[2 * qi + 1];
            const v3 t_nrm = V3(ta.w, tb.x, tb.y);
            const float dist_sqr = (float)(dx * dx + dy * dy);
            const float dist = sqrtf(dist_sqr);
            const float w_xy = mrf_exp(-dist_sqr / (2.0f * variance));
            const float w_normal = mrf_pow2k(fminf(fmaxf(dot(t_nrm, c_nrm), MR_BIL_EPS), 1.0f), 7);
            const float w_depth = mrf_exp(-(fabsf(tb.z - c_z) / fmaxf((MODE == 2 ? tb.w : c_dz) * dist, MR_BIL_EPS)));
            const float w = w_xy * w_normal * w_depth;
            if (MODE == 2) acc = acc + V3(grad_out4[4 * qi], grad_out4[4 * qi + 1], grad_out4[4 * qi + 2]) * w;
            else { acc = acc + V3(ta.x, ta.y, ta.z) * w; accum_w += w; }
        }
    }
    if (MODE == 0) { out[4 * pi] = acc.x; out[4 * pi + 1] = acc.y; out[4 * pi + 2] = acc.z; out[4 * pi + 3] = fmaxf(accum_w, MR_BIL_EPS); }
    else if (MODE == 1) { const float d = fmaxf(accum_w, MR_BIL_EPS); out[3 * pi] = acc.x / d; out[3 * pi + 1] = acc.y / d; out[3 * pi + 2] = acc.z / d; }
    else { out[3 * pi] = acc.x; out[3 * pi + 1] = acc.y; out[3 * pi + 2] = acc.z; }
}

}  // namespace mr

using namespace mr;

namespace mr {
// The five buffers the frame loop denoises share normal / depth guides, i.e. the weights: one pass over the window accumulates all of them.
// Taps are read from one packed 80-byte record per pixel {n.xyz z | dz c0 | c1 c2.x | c2.yz c3.xy | c3.z c4} (5 x 16-byte loads instead of
// 16 scalar ones), and the terms that depend only on the tap offset — dist = sqrt(dx^2 + dy^2) and exp(-d^2 / 2 sigma^2) — come from an LDS
// table filled once per workgroup with the very expressions of the per-tap code (same bits). out_k = sum w col_k / max(sum w, 1e-4).
struct Bil5 { const float* col[5]; float* out[5]; };
__global__ void __launch_bounds__(MR_BLOCK) k_bilateral_pack5(size_t n, const float* __restrict__ nrm, const float* __restrict__ zdz, Bil5 P, float4* __restrict__ rec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v3 nn = ld3(nrm, i);
    const float len = sqrtf(fmaxf(dot(nn, nn), 1e-20f));
    v3 c[5];
#pragma unroll
    for (int k = 0; k < 5; k++) c[k] = ld3(P.col[k], i);
    float4 r[5];
    r[0].x = nn.x / len; r[0].y = nn.y / len; r[0].z = nn.z / len; r[0].w = zdz[2 * i];
    r[1].x = zdz[2 * i + 1]; r[1].y = c[0].x; r[1].z = c[0].y; r[1].w = c[0].z;
    r[2].x = c[1].x; r[2].y = c[1].y; r[2].z = c[1].z; r[2].w = c[2].x;
    r[3].x = c[2].y; r[3].y = c[2].z; r[3].z = c[3].x; r[3].w = c[3].y;
    r[4].x = c[3].z; r[4].y = c[4].x; r[4].z = c[4].y; r[4].w = c[4].z;
#pragma unroll
    for (int k = 0; k < 5; k++) rec[5 * i + k] = r[k];
}
#define MR_BIL_MAXRAD 21     // sigma = 4 (factor 2, renderer_restir.py:530); larger windows fall back to the per-buffer kernel
__global__ void __launch_bounds__(MR_BLOCK) k_bilateral5(int fx, int fy, float sigma, const float4* __restrict__ rec, Bil5 P) {
    __shared__ float s_wxy[(2 * MR_BIL_MAXRAD + 1) * (2 * MR_BIL_MAXRAD + 1)], s_dist[(2 * MR_BIL_MAXRAD + 1) * (2 * MR_BIL_MAXRAD + 1)];
    const float variance = sigma * sigma;
    const int rad = 2 * (int)ceilf(sigma * 2.5f) + 1, wdt = 2 * rad + 1;
    for (int i = threadIdx.x; i < wdt * wdt; i += MR_BLOCK) {
        const int dy = i / wdt - rad, dx = i % wdt - rad;
        const float dist_sqr = (float)(dx * dx + dy * dy);
        s_dist[i] = sqrtf(dist_sqr);
        s_wxy[i] = mrf_exp(-dist_sqr / (2.0f * variance));
    }
    __syncthreads();
    const int tiles_x = (fx + 15) >> 4;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 16 + (int)(threadIdx.x & 15), y = ty * 16 + (int)(threadIdx.x >> 4);
    if (x >= fx || y >= fy) return;
    const size_t pi = (size_t)y * fx + x;
    const float4 ca = rec[5 * pi];
    const v3 c_nrm = V3(ca.x, ca.y, ca.z);
    const float c_z = ca.w, c_dz = rec[5 * pi + 1].x;
    float accum_w = 0.f; v3 acc[5];
#pragma unroll
    for (int k = 0; k < 5; k++) acc[k] = V3(0.f);
    for (int dy = -rad; dy <= rad; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= fy) continue;
        for (int dx = -rad; dx <= rad; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= fx) continue;
            const float4* __restrict__ t = rec + 5 * ((size_t)yy * fx + xx);
            const float4 t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4];
            const int li = (dy + rad) * wdt + (dx + rad);
            const float w_normal = mrf_pow2k(fminf(fmaxf(dot(V3(t0.x, t0.y, t0.z), c_nrm), MR_BIL_EPS), 1.0f), 7);
            const float w_depth = mrf_exp(-(fabsf(t0.w - c_z) / fmaxf(c_dz * s_dist[li], MR_BIL_EPS)));
            const float w = s_wxy[li] * w_normal * w_depth;
            acc[0] = acc[0] + V3(t1.y, t1.z, t1.w) * w; acc[1] = acc[1] + V3(t2.x, t2.y, t2.z) * w; acc[2] = acc[2] + V3(t2.w, t3.x, t3.y) * w;
            acc[3] = acc[3] + V3(t3.z, t3.w, t4.x) * w; acc[4] = acc[4] + V3(t4.y, t4.z, t4.w) * w;
            accum_w += w;
        }
    }
    const float d = fmaxf(accum_w, MR_BIL_EPS);
#pragma unroll
    for (int k = 0; k < 5; k++) { P.out[k][3 * pi] = acc[k].x / d; P.out[k][3 * pi + 1] = acc[k].y / d; P.out[k][3 * pi + 2] = acc[k].z / d; }
}
int launch_bilateral_divided(int fx, int fy, float sigma, const float* col, const float* nrm, const float* zdz, float* scratch, float* out3, hipStream_t s);
// scratch: f32[N, 20]
int launch_bilateral5(int fx, int fy, float sigma, const float* const col[5], const float* nrm, const float* zdz, float* scratch, float* const out[5], hipStream_t s) {
    const size_t n = (size_t)fx * fy;
    if (2 * (int)ceilf(sigma * 2.5f) + 1 > MR_BIL_MAXRAD) {
        for (int k = 0; k < 5; k++) { int rc = launch_bilateral_divided(fx, fy, sigma, col[k], nrm, zdz, scratch, out[k], s); if (rc) return rc; }
        return 0;
    }
    Bil5 P; for (int k = 0; k < 5; k++) { P.col[k] = col[k]; P.out[k] = out[k]; }
    k_bilateral_pack5<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, s>>>(n, nrm, zdz, P, reinterpret_cast<float4*>(scratch));
    k_bilateral5<<<((fx + 15) / 16) * ((fy + 15) / 16), MR_BLOCK, 0, s>>>(fx, fy, sigma, reinterpret_cast<const float4*>(scratch), P);
    MR_LAUNCH_CHECK("bilateral5");
    return 0;
}
int launch_bilateral_divided(int fx, int fy, float sigma, const float* col, const float* nrm, const float* zdz, float* scratch, float* out3, hipStream_t s) {
    const size_t n = (size_t)fx * fy;
    k_bilateral_pack<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, s>>>(n, col, nrm, zdz, reinterpret_cast<float4*>(scratch));
    k_bilateral<1><<<((fx + 15) / 16) * ((fy + 15) / 16), MR_BLOCK, 0, s>>>(fx, fy, sigma, reinterpret_cast<const float4*>(scratch), nullptr, out3);
    MR_LAUNCH_CHECK("bilateral_divided");
    return 0;
}
// process_normal_ao (EAWDenoise.slang:591-651): 8 x 8 window (offsets -4 .. 3 in x and y) of foreground neighbours, mean of clamp(n_q . n_p, 0, 1),
// weight = clamp(50 (1 - mean), 0, 1) splat to three channels; background pixels get 0.  The sum runs x-offset outer, y-offset inner as written there.
__global__ void __launch_bounds__(MR_BLOCK) k_normal_ao(int fx, int fy, const float* __restrict__ occ, const float* __restrict__ normal, float* __restrict__ out) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= fx * fy) return;
    float v = 0.f;
    if (!(occ[pi] < 0.1f)) {
        const int x = pi % fx, y = pi / fx;
        const v3 nv = ld3(normal, pi);
        float sum = 0.f; int count = 0;
        for (int i = -4; i < 4; i++)
            for (int j = -4; j < 4; j++) {
                const int ux = x + i, uy = y + j;
                if (ux < 0 || uy < 0 || ux >= fx || uy >= fy) continue;
                const int q = uy * fx + ux;
                if (occ[q] < 0.1f) continue;
                sum += fminf(1.0f, fmaxf(dot(ld3(normal, q), nv), 0.0f));
                count++;
            }
        const float w = 1.f - sum / (float)count;
        v = fminf(fmaxf(w * 50.f, 0.f), 1.f);
    }
    st3(out, pi, V3(v));
}

}  // namespace mr

extern "C" {

int mirres_normal_ao(int fx, int fy, const float* occ, const float* normal, float* out_ao, void* stream) {
    if (fx <= 0 || fy <= 0 || !occ || !normal || !out_ao) { set_error("mirres_normal_ao: bad argument"); return MIRRES_E_ARG; }
    k_normal_ao<<<grid_for((size_t)fx * fy, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(fx, fy, occ, normal, out_ao);
    MR_LAUNCH_CHECK("normal_ao");
    return MIRRES_OK;
}

int mirres_eaw(int fx, int fy, int step_width, float c_phi, float n_phi, float p_phi, const float* occ, const float* color, const float* normal,
               const float* pos, float* out, void* stream) {
    if (fx <= 0 || fy <= 0 || !occ || !color || !normal || !pos || !out || color == out) { set_error("mirres_eaw: bad argument"); return MIRRES_E_ARG; }
    k_eaw<<<grid_for((size_t)fx * fy, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(fx, fy, step_width, c_phi, n_phi, p_phi, occ, color, normal, pos, out);
    MR_LAUNCH_CHECK("eaw");
    return MIRRES_OK;
}

int mirres_eaw_bwd_gather(int fx, int fy, int step_width, float c_phi, float n_phi, float p_phi, const float* occ, const float* color, const float* normal,
                          const float* pos, const float* grad_out, float* scratch4, float* g_color, float* g_normal, float* g_pos, void* stream) {
    if (fx <= 0 || fy <= 0 || !occ || !color || !normal || !pos || !grad_out || !scratch4 || !g_color) { set_error("mirres_eaw_bwd_gather: bad argument"); return MIRRES_E_ARG; }
    const int grd = grid_for((size_t)fx * fy, MR_BLOCK);
    k_eaw_bwd_sums<<<grd, MR_BLOCK, 0, (hipStream_t)stream>>>(fx, fy, step_width, c_phi, n_phi, p_phi, occ, color, normal, pos, grad_out, reinterpret_cast<float4*>(scratch4));
    k_eaw_bwd_gather<<<grd, MR_BLOCK, 0, (hipStream_t)stream>>>(fx, fy, step_width, c_phi, n_phi, p_phi, occ, color, normal, pos, grad_out,
                                                                reinterpret_cast<const float4*>(scratch4), g_color, g_normal, g_pos);
    MR_LAUNCH_CHECK("eaw_bwd_gather");
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


/* bilateral_denoiser_fwd / _bwd (nerf/renderutils/ops.py:173-188, c_src/denoising.cu). col f32[N,3], nrm f32[N,3] (normalised here), zdz f32[N,2],
 * out f32[N,4] = (sum w col, max(sum w, 1e-4)); scratch f32[N,8] holds the packed taps.                                                    */
int mirres_bilateral(int fx, int fy, float sigma, const float* col, const float* nrm, const float* zdz, float* scratch, float* out4, void* stream) {
    if (fx <= 0 || fy <= 0 || !(sigma > 0.f) || !col || !nrm || !zdz || !scratch || !out4) { set_error("mirres_bilateral: bad argument"); return MIRRES_E_ARG; }
    const size_t n = (size_t)fx * fy; hipStream_t s = (hipStream_t)stream;
    k_bilateral_pack<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, s>>>(n, col, nrm, zdz, reinterpret_cast<float4*>(scratch));
    k_bilateral<0><<<((fx + 15) / 16) * ((fy + 15) / 16), MR_BLOCK, 0, s>>>(fx, fy, sigma, reinterpret_cast<const float4*>(scratch), nullptr, out4);
    MR_LAUNCH_CHECK("bilateral");
    return MIRRES_OK;
}
int mirres_bilateral_bwd(int fx, int fy, float sigma, const float* nrm, const float* zdz, const float* grad_out4, float* scratch, float* col_grad, void* stream) {
    if (fx <= 0 || fy <= 0 || !(sigma > 0.f) || !nrm || !zdz || !grad_out4 || !scratch || !col_grad) { set_error("mirres_bilateral_bwd: bad argument"); return MIRRES_E_ARG; }
    const size_t n = (size_t)fx * fy; hipStream_t s = (hipStream_t)stream;
    k_bilateral_pack<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, s>>>(n, nullptr, nrm, zdz, reinterpret_cast<float4*>(scratch));
    k_bilateral<2><<<((fx + 15) / 16) * ((fy + 15) / 16), MR_BLOCK, 0, s>>>(fx, fy, sigma, reinterpret_cast<const float4*>(scratch), grad_out4, col_grad);
    MR_LAUNCH_CHECK("bilateral_bwd");
    return MIRRES_OK;
}
}  // extern "C"
