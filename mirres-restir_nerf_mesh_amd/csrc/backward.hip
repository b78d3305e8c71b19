// backward.hip — adjoints that the reference obtains from Slang autodiff / torch autograd, written by hand for gfx950:
//   * mirres_final_shading_bwd : process_FinalShading.bwd (Resampling.py:179-214) — per-pixel forward-mode Jacobian
//     (dual numbers, 3 partials per sweep over the input groups normal / kd / (roughness, metallic) / Li) contracted with
//     the incoming cotangents. Same branch selection as the forward kernel, so it is the derivative of exactly what ran.
//   * mirres_matnet_bwd        : backward of MLPTexture3D.sample (render_helper.py:93-104): fp32 MLP weight gradients
//     (block-level LDS reduction, then one atomic per weight per block) and hash-grid gradients scattered with fp32 atomics
//     into the fp32 master table (tcnn accumulates its grid gradient the same way).
#ifndef MR_LEAN_FP
#define MR_LEAN_FP 1      // device_math.hpp: short division / square-root sequences (same bits as the compiler's for the renderer's operand range)
#endif
#include "engine.hpp"
#include "device_math.hpp"
#include "device_light.hpp"
#include "device_grid.hpp"
#include <hip/hip_fp16.h>
#include <cmath>

namespace mr {

#define MR_BLOCK 256

// ---------------------------------------------------------------- dual numbers
template <int NP> struct Dual { float v; float d[NP]; };
template <int NP> MR_DEV Dual<NP> mk(float v) { Dual<NP> r; r.v = v;
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = 0.f; return r; }
template <int NP> MR_DEV Dual<NP> operator+(Dual<NP> a, Dual<NP> b) { Dual<NP> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int NP> MR_DEV Dual<NP> operator-(Dual<NP> a, Dual<NP> b) { Dual<NP> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int NP> MR_DEV Dual<NP> operator-(Dual<NP> a) { Dual<NP> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = -a.d[i]; return r; }
template <int NP> MR_DEV Dual<NP> operator*(Dual<NP> a, Dual<NP> b) { Dual<NP> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int NP> MR_DEV Dual<NP> operator*(Dual<NP> a, float s) { Dual<NP> r; r.v = a.v * s;
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = a.d[i] * s; return r; }
template <int NP> MR_DEV Dual<NP> operator*(float s, Dual<NP> a) { return a * s; }
template <int NP> MR_DEV Dual<NP> operator+(Dual<NP> a, float s) { a.v += s; return a; }
template <int NP> MR_DEV Dual<NP> operator+(float s, Dual<NP> a) { a.v += s; return a; }
template <int NP> MR_DEV Dual<NP> operator-(float s, Dual<NP> a) { return mk<NP>(s) - a; }
template <int NP> MR_DEV Dual<NP> operator-(Dual<NP> a, float s) { a.v -= s; return a; }
template <int NP> MR_DEV Dual<NP> operator/(Dual<NP> a, Dual<NP> b) { Dual<NP> r; r.v = mr_div(a.v, b.v); float ib = mr_rcp(b.v);
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int NP> MR_DEV Dual<NP> operator/(float s, Dual<NP> b) { return mk<NP>(s) / b; }
template <int NP> MR_DEV Dual<NP> operator/(Dual<NP> a, float s) { return a * mr_rcp(s); }
template <int NP> MR_DEV Dual<NP> dsqrt(Dual<NP> a) { Dual<NP> r; r.v = mr_sqrt(a.v); float k = a.v > 0.f ? mr_div(0.5f, r.v) : 0.f;
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = a.d[i] * k; return r; }
template <int NP> MR_DEV Dual<NP> dmax0(Dual<NP> a) { return a.v > 0.f ? a : mk<NP>(0.f); }  // max(a, 0)
template <int NP> MR_DEV Dual<NP> dpow5(Dual<NP> a) { Dual<NP> r; float a2 = a.v * a.v; r.v = a2 * a2 * a.v; float k = 5.f * a2 * a2;
#pragma unroll
    for (int i = 0; i < NP; i++) r.d[i] = a.d[i] * k; return r; }

template <int NP> struct D3 { Dual<NP> x, y, z; };
template <int NP> MR_DEV D3<NP> mk3(v3 v) { D3<NP> r; r.x = mk<NP>(v.x); r.y = mk<NP>(v.y); r.z = mk<NP>(v.z); return r; }
template <int NP> MR_DEV Dual<NP> ddot(D3<NP> a, D3<NP> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
template <int NP> MR_DEV D3<NP> operator+(D3<NP> a, D3<NP> b) { D3<NP> r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
template <int NP> MR_DEV D3<NP> operator*(D3<NP> a, Dual<NP> s) { D3<NP> r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
template <int NP> MR_DEV D3<NP> operator*(D3<NP> a, D3<NP> b) { D3<NP> r; r.x = a.x * b.x; r.y = a.y * b.y; r.z = a.z * b.z; return r; }

template <int NP> MR_DEV Dual<NP> d_lambda(Dual<NP> a2, Dual<NP> c) {  // evalLambdaGGX brdfDi.slang:40-47
    if (c.v <= 0) return mk<NP>(0.f);
    Dual<NP> c2 = c * c;
    Dual<NP> tan2 = dmax0(1.0f - c2) / c2;
    return 0.5f * (-1.0f + dsqrt(1.0f + a2 * tan2));
}

// process_FinalShading (FinalShading.slang:14-109) on dual numbers. GROUP selects which inputs carry unit partials:
// 0 normal, 1 kd, 2 (roughness, metallic, -), 3 Li.
template <int GROUP>
MR_DEV void shade_dual(v3 n_, v3 rd, v3 kd_, float rough_, float metal_, v3 dir, v3 Li_, D3<3>& color, D3<3>& dl, D3<3>& sl) {
    constexpr int NP = 3;
    D3<NP> n = mk3<NP>(n_), kd = mk3<NP>(kd_), Li = mk3<NP>(Li_);
    Dual<NP> rough = mk<NP>(rough_), metal = mk<NP>(metal_);
    if (GROUP == 0) { n.x.d[0] = 1.f; n.y.d[1] = 1.f; n.z.d[2] = 1.f; }
    if (GROUP == 1) { kd.x.d[0] = 1.f; kd.y.d[1] = 1.f; kd.z.d[2] = 1.f; }
    if (GROUP == 2) { rough.d[0] = 1.f; metal.d[1] = 1.f; }
    if (GROUP == 3) { Li.x.d[0] = 1.f; Li.y.d[1] = 1.f; Li.z.d[2] = 1.f; }
    // create_frame (helperDi.slang:18-28)
    const float sign = (n.z.v > 0) ? 1.0f : -1.0f;
    Dual<NP> a = -1.0f / (sign + n.z);
    Dual<NP> b = n.x * n.y * a;
    D3<NP> fx, fy;
    fx.x = 1.0f + sign * (n.x * n.x * a); fx.y = sign * b; fx.z = -sign * n.x;
    fy.x = b; fy.y = sign + n.y * n.y * a; fy.z = -n.y;
    D3<NP> mrd = mk3<NP>(-rd), dd = mk3<NP>(dir);
    D3<NP> wi, wo;
    wi.x = ddot(fx, mrd); wi.y = ddot(fy, mrd); wi.z = ddot(n, mrd);
    wo.x = ddot(fx, dd); wo.y = ddot(fy, dd); wo.z = ddot(n, dd);
    // lobes (only gate the terms)
    D3<NP> specular;
    specular.x = 0.04f * (1.0f - metal) + kd.x * metal; specular.y = 0.04f * (1.0f - metal) + kd.y * metal; specular.z = 0.04f * (1.0f - metal) + kd.z * metal;
    Dual<NP> alpha = rough * rough;
    if (alpha.v < 0.01f * 0.01f) alpha = mk<NP>(0.f);
    float pD = luminance(kd_) * (1.f - metal_);
    float cosv = dot(-rd, n_);
    float p5 = mrf_pow5(fmaxf(1 - cosv, 0));
    v3 sp_ = V3(0.04f) * (1.0f - metal_) + kd_ * metal_;
    float pS = luminance(V3(sp_.x + (1 - sp_.x) * p5, sp_.y + (1 - sp_.y) * p5, sp_.z + (1 - sp_.z) * p5)) * (metal_ + (1.f - metal_));
    D3<NP> dv = mk3<NP>(V3(0.f)), sv = mk3<NP>(V3(0.f));
    const bool low = fminf(wi.z.v, wo.z.v) < 1e-6f;
    if (pD > 0.f && !low) {
        Dual<NP> f = wo.z * 0.31830988f;
        if (!(f.v > 0.f)) f = mk<NP>(fmaxf(f.v, 0.f));
        dv = Li * f;
    }
    if (pS > 0.f && !low && alpha.v != 0.f) {
        D3<NP> hs = wi + wo;
        Dual<NP> inv = 1.0f / dsqrt(ddot(hs, hs));
        D3<NP> h = hs * inv;
        Dual<NP> woDotH = ddot(wi, h);
        Dual<NP> a2 = alpha * alpha;
        Dual<NP> dd_ = ((h.z * a2 - h.z) * h.z + 1.0f);
        Dual<NP> Dn = a2 / (dd_ * dd_ * 3.141592653589793f);
        Dual<NP> G = 1.0f / (1.0f + d_lambda(a2, wi.z) + d_lambda(a2, wo.z));
        Dual<NP> pw = dpow5(dmax0(1.0f - woDotH));
        D3<NP> F;
        F.x = specular.x + (1.0f - specular.x) * pw; F.y = specular.y + (1.0f - specular.y) * pw; F.z = specular.z + (1.0f - specular.z) * pw;
        Dual<NP> k = Dn * G * 0.25f / wi.z;
        sv = (F * k) * Li;
    }
    Dual<NP> om = 1.0f - metal;
    color.x = kd.x * om * dv.x + sv.x; color.y = kd.y * om * dv.y + sv.y; color.z = kd.z * om * dv.z + sv.z;
    dl = dv; sl = sv;
}

template <int GROUP>
MR_DEV v3 contract(v3 n, v3 rd, v3 kd, float rough, float metal, v3 dir, v3 Li, v3 gc, v3 gd, v3 gs) {
    D3<3> c, dl, sl;
    shade_dual<GROUP>(n, rd, kd, rough, metal, dir, Li, c, dl, sl);
    float o[3];
#pragma unroll
    for (int j = 0; j < 3; j++)
        o[j] = (gc.x * c.x.d[j] + gc.y * c.y.d[j] + gc.z * c.z.d[j]) + (gd.x * dl.x.d[j] + gd.y * dl.y.d[j] + gd.z * dl.z.d[j]) +
               (gs.x * sl.x.d[j] + gs.y * sl.y.d[j] + gs.z * sl.z.d[j]);
    return V3(o[0], o[1], o[2]);
}

__global__ void __launch_bounds__(MR_BLOCK) k_final_shading_bwd(int N, const float* __restrict__ occ, const float* __restrict__ normal,
                                                                const float* __restrict__ ray_dir, const float* __restrict__ kd, const float* __restrict__ rm,
                                                                const float* __restrict__ fdir, const float* __restrict__ fdist, const float* __restrict__ fLi,
                                                                const float* __restrict__ g_color, const float* __restrict__ g_diff, const float* __restrict__ g_spec,
                                                                float* __restrict__ g_normal, float* __restrict__ g_kd, float* __restrict__ g_rm, float* __restrict__ g_Li) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= N) return;
    v3 gn = V3(0.f), gk = V3(0.f), gr = V3(0.f), gl = V3(0.f);
    if (occ[pi] > 0.1f && fdist[pi] > 0.f) {
        const v3 n = ld3(normal, pi), rd = ld3(ray_dir, pi), k = ld3(kd, pi), dir = ld3(fdir, pi), Li = ld3(fLi, pi);
        const float rough = rm[2 * (size_t)pi], metal = rm[2 * (size_t)pi + 1];
        const v3 gc = ld3(g_color, pi), gd = ld3(g_diff, pi), gs = ld3(g_spec, pi);
        if (g_normal) gn = contract<0>(n, rd, k, rough, metal, dir, Li, gc, gd, gs);
        if (g_kd) gk = contract<1>(n, rd, k, rough, metal, dir, Li, gc, gd, gs);
        if (g_rm) gr = contract<2>(n, rd, k, rough, metal, dir, Li, gc, gd, gs);
        if (g_Li) gl = contract<3>(n, rd, k, rough, metal, dir, Li, gc, gd, gs);
    }
    if (g_normal) st3(g_normal, pi, gn);
    if (g_kd) st3(g_kd, pi, gk);
    if (g_rm) { g_rm[2 * (size_t)pi] = gr.x; g_rm[2 * (size_t)pi + 1] = gr.y; }
    if (g_Li) st3(g_Li, pi, gl);
}

// ---------------------------------------------------------------- fused backward of the frame's direct-lighting sums (mirres_render_bwd)
// total = sum over samples s of FinalShading(normal, kd, rm, Li_s) with Li_s = W_s * env(dir_s) * [visible_s] (EvaluateFinalSamples_di): one thread
// per pixel walks the tape of the forward call, re-forms each sample's (dir, Li) exactly as the forward did, contracts the dual-number
// Jacobian of the shading with the cotangents of the three sums (the same code as k_final_shading_bwd) and scatters d/dLi into the four
// environment texels of the bilinear lookup (k_eval_final_bwd). tex is the flipped map the forward sampled; g_env is in the caller's layout.
// MR_DBW_SPLIT lanes share a pixel, each walking every MR_DBW_SPLIT-th sample of the tape (one thread per pixel left the chip with ten waves per
// SIMD in total, each a serial loop over all samples: 5.5 ms for 640 k pixels x 32 samples at 10 % VALU utilisation); the per-pixel gradients
// are then summed across the lanes by shuffles (a fixed tree: deterministic), the environment scatter is atomic as before.
// The environment gradient is a scatter onto few texels (importance sampling sends most samples to the brightest ones) and global atomics onto
// one address serialise (~88 per microsecond): 245 M adds took 5.5 ms. Every workgroup therefore first accumulates into an LDS table keyed by
// texel (open addressing, two probes; an LDS same-address add costs cycles, not a memory round trip) and flushes each occupied entry with one
// global atomic per channel at the end; an insert that finds both probes taken by other texels goes to global memory directly.
#define MR_DBW_SPLIT 8
#define MR_DBW_TABLE 2048
#define MR_DBW_LIST 2048
// Round 6 (profiles/r06_pmc_train.txt, r06_atomic_rate.txt): the kernel ran at the memory side's atomic REQUEST rate — 65 M requests per launch at 17 G/s; MI355X takes
// 21 G scattered fp32 atomic requests per second whatever the scope (every device-scope atomic leaves the XCD's L2: TCC_EA0_ATOMIC == TCC_ATOMIC), but the lanes of ONE
// instruction that fall into the same 32-byte sector travel as one request (G consecutive lanes on G consecutive dwords: G x 21 G lane-atomics/s). A texel's three
// channels are 12 consecutive bytes. So no contribution goes to global memory from the sample loop any more: what the hash table cannot hold (both probes taken by other
// texels: a flat map spreads a workgroup's 4096 texel updates over as many texels) is appended to an LDS list, and table and list are flushed at the end by FOUR
// lanes per entry (three channels + an idle lane) — a third of the requests. A full list (cannot happen: 2048 + 2048 entries per workgroup of 4096 updates) still falls
// back to direct atomics.
struct EnvScatter { int* keys; float* vals; int* lkeys; float* lvals; int* lcount; };
MR_DEV void env_grad_add(const EnvScatter& S, float* g_env, int texel, v3 g) {
    uint32_t h = ((uint32_t)texel * 2654435761u) >> (32 - 11);
#pragma unroll
    for (int probe = 0; probe < 2; probe++) {
        const int old = atomicCAS(&S.keys[h], -1, texel);
        if (old == -1 || old == texel) { atomicAdd(&S.vals[3 * h], g.x); atomicAdd(&S.vals[3 * h + 1], g.y); atomicAdd(&S.vals[3 * h + 2], g.z); return; }
        h = (h + 1) & (MR_DBW_TABLE - 1);
    }
    const int at = atomicAdd(S.lcount, 1);
    if (at < MR_DBW_LIST) { S.lkeys[at] = texel; S.lvals[3 * at] = g.x; S.lvals[3 * at + 1] = g.y; S.lvals[3 * at + 2] = g.z; return; }
    float* dst = g_env + 3 * (size_t)texel;
    atomicAdd(dst, g.x); atomicAdd(dst + 1, g.y); atomicAdd(dst + 2, g.z);
}
// entries [0, n) of (keys, vals) -> global memory, four lanes per entry: lanes 4 e .. 4 e + 2 add the three channels of one texel in ONE instruction
MR_DEV void env_grad_flush(const int* keys, const float* vals, int n, float* g_env) {
    for (int j = threadIdx.x; j < 4 * n; j += MR_BLOCK) {
        const int e = j >> 2, c = j & 3;
        const int key = keys[e];
        if (c < 3 && key >= 0) atomicAdd(g_env + 3 * (size_t)key + c, vals[3 * e + c]);
    }
}
__global__ void __launch_bounds__(MR_BLOCK) k_direct_bwd(EnvD E, int N, int S, const float* __restrict__ occ, const float* __restrict__ normal,
                                                         const float* __restrict__ ray_dir_raw, const float* __restrict__ kd, const float* __restrict__ rm,
                                                         const float4* __restrict__ tape, const float* __restrict__ g_color, const float* __restrict__ g_diff,
                                                         const float* __restrict__ g_spec, float* __restrict__ g_normal, float* __restrict__ g_kd,
                                                         float* __restrict__ g_rm, float* __restrict__ g_env) {
    __shared__ int s_keys[MR_DBW_TABLE];
    __shared__ float s_vals[3 * MR_DBW_TABLE];
    __shared__ int s_lkeys[MR_DBW_LIST];
    __shared__ float s_lvals[3 * MR_DBW_LIST];
    __shared__ int s_lcount;
    for (int i = threadIdx.x; i < MR_DBW_TABLE; i += MR_BLOCK) { s_keys[i] = -1; s_vals[3 * i] = 0.f; s_vals[3 * i + 1] = 0.f; s_vals[3 * i + 2] = 0.f; }
    if (threadIdx.x == 0) s_lcount = 0;
    __syncthreads();
    const EnvScatter ES = {s_keys, s_vals, s_lkeys, s_lvals, &s_lcount};
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // lanes l, l + 8, l + 16, ... of a wave hold the same sample phase of consecutive pixels: 8 consecutive tape records per phase
    const int wave_px = (int)(t >> 6) * (64 / MR_DBW_SPLIT);
    const int lane = threadIdx.x & 63;
    const int pi = wave_px + (lane & (64 / MR_DBW_SPLIT - 1)), sub = lane / (64 / MR_DBW_SPLIT);
    const bool live = pi < N;
    v3 gn = V3(0.f), gk = V3(0.f), gr = V3(0.f);
    if (live && occ[pi] > 0.1f) {
        const v3 n = ld3(normal, pi), k = ld3(kd, pi);
        v3 rd = ld3(ray_dir_raw, pi);
        { const float l = fmaxf(mr_sqrt(dot(rd, rd)), 1e-6f); rd = V3(mr_div(rd.x, l), mr_div(rd.y, l), mr_div(rd.z, l)); }   // the forward's k_prep (F.normalize, eps 1e-6)
        const float rough = rm[2 * (size_t)pi], metal = rm[2 * (size_t)pi + 1];
        const v3 gc = ld3(g_color, pi), gd = ld3(g_diff, pi), gs = ld3(g_spec, pi);
        for (int s = sub; s < S; s += MR_DBW_SPLIT) {
            const float4 a = tape[2 * ((size_t)s * N + pi)], b = tape[2 * ((size_t)s * N + pi) + 1];
            if (!(a.x > 0.1f) || !(b.z > 0.f)) continue;                  // empty reservoir or occluded: Li = 0, dist = 0 -> no contribution
            const v3 dir = oct_decode(V2(a.y, a.z));
            const v3 Li = b.y * env_radiance(E, dir);
            gn = gn + contract<0>(n, rd, k, rough, metal, dir, Li, gc, gd, gs);
            gk = gk + contract<1>(n, rd, k, rough, metal, dir, Li, gc, gd, gs);
            gr = gr + contract<2>(n, rd, k, rough, metal, dir, Li, gc, gd, gs);
            if (g_env) {
                const v3 gl = contract<3>(n, rd, k, rough, metal, dir, Li, gc, gd, gs) * b.y;    // d/d env = W * d/dLi
                int idx[4]; float w[4];
                if (env_le_footprint(ngp_dir(dir), E.W, E.H, idx, w)) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int ty = idx[q] / E.W, tx = idx[q] - ty * E.W;
                        env_grad_add(ES, g_env, (E.H - 1 - ty) * E.W + tx, V3(gl.x * w[q], gl.y * w[q], gl.z * w[q]));   // tex row ty = caller's row H-1-ty (k_flip_env)
                    }
                }
            }
        }
    }
    // sum over the sample phases: lanes that differ in bits 3..5
    float v[8] = {gn.x, gn.y, gn.z, gk.x, gk.y, gk.z, gr.x, gr.y};
#pragma unroll
    for (int off = 64 / MR_DBW_SPLIT; off < 64; off <<= 1) {
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] += __shfl_xor(v[q], off, 64);
    }
    if (g_env) {
        __syncthreads();
        env_grad_flush(s_keys, s_vals, MR_DBW_TABLE, g_env);
        env_grad_flush(s_lkeys, s_lvals, s_lcount < MR_DBW_LIST ? s_lcount : MR_DBW_LIST, g_env);
    }
    if (live && sub == 0) {
        if (g_normal) st3(g_normal, pi, V3(v[0], v[1], v[2]));
        if (g_kd) st3(g_kd, pi, V3(v[3], v[4], v[5]));
        if (g_rm) { g_rm[2 * (size_t)pi] = v[6]; g_rm[2 * (size_t)pi + 1] = v[7]; }
    }
}
int launch_direct_bwd(const float* tex, int Wc, int Hc, int N, int S, const float* occ, const float* normal, const float* ray_dir_raw, const float* kd, const float* rm,
                      const float* tape, const float* g_color, const float* g_diff, const float* g_spec, float* g_normal, float* g_kd, float* g_rm, float* g_env,
                      hipStream_t s) {
    EnvD E; E.tex = tex; E.W = Wc; E.H = Hc; E.pdf = nullptr; E.cdf = nullptr; E.mpdf = nullptr; E.mcdf = nullptr;
    k_direct_bwd<<<grid_for((size_t)N * MR_DBW_SPLIT, MR_BLOCK), MR_BLOCK, 0, s>>>(E, N, S, occ, normal, ray_dir_raw, kd, rm, reinterpret_cast<const float4*>(tape), g_color, g_diff, g_spec,
                                                            g_normal, g_kd, g_rm, g_env);
    MR_LAUNCH_CHECK("direct_bwd");
    return 0;
}

// ---------------------------------------------------------------- material-field backward (level table and index arithmetic: device_grid.hpp)
struct MatNetB { const __half2* grid; const float *w0, *w1, *w2; float aabb_min[3], aabb_max[3], mn[6], mx[6]; };

// Weight gradients are outer-product sums over the points (gW1[o][k] = sum_p gh2_p[o] * h1_p[k], ...). The first version added every
// product to LDS with an atomic — 2 240 LDS atomics per point onto 2 240 addresses shared by the whole workgroup, 15 of the kernel's 17.6 ms.
// Now each wave stages its 64 points' two factor vectors in LDS and every lane accumulates a fixed set of matrix entries in registers over
// all the tiles its wave processes (lane l owns entries l, l + 64, ...: the k index is l % 32 for all of them, so one factor is read once per
// point); one LDS reduction over the four waves and one global atomic per entry per workgroup at the end.
#define MR_BW_WAVES (MR_BLOCK / 64)
#define MR_BW_LD 33     // padded row of the staged [64][32] factors
#define MR_BW_TABLE_LOG2 10
#define MR_BW_TABLE (1 << MR_BW_TABLE_LOG2)   // per-wave aggregation table of the coarse grid levels: keys in U (4 KB of 8.4), values in V (8 KB of 8.4)
#define MR_BW_COARSE 8
typedef GridLevels GridLevelsB;
static GridLevelsB host_levels_b() { return host_levels(nullptr); }
// Two waves per SIMD (256 registers, the rest in scratch) instead of one (512 + AGPRs): -4 % on the kernel (profiles/r06_ab_matnet_bwd_builds.txt). Forming every layer's
// weight-gradient outer product right after its adjoint (shorter live ranges in the source) was measured too: the compiler hoists the LDS loads of the fully unrolled
// mat-vecs either way — +6 % slower with one wave, +-0 with two; not kept.
#ifndef MR_BW_MINWAVES
#define MR_BW_MINWAVES 2
#endif
__global__ void __launch_bounds__(MR_BLOCK, MR_BW_MINWAVES) k_matnet_bwd(MatNetB M, GridLevelsB L, const float* __restrict__ pos, int n, const float* __restrict__ gout,
                                                         float* __restrict__ g_params, float* __restrict__ g_w0, float* __restrict__ g_w1, float* __restrict__ g_w2,
                                                         float* __restrict__ g_pos) {
    __shared__ float sw0[1024], sw1[1024], sw2[192];
    __shared__ float sU[MR_BW_WAVES][64 * MR_BW_LD], sV[MR_BW_WAVES][64 * MR_BW_LD];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) { sw0[i] = M.w0[i]; sw1[i] = M.w1[i]; }
    for (int i = threadIdx.x; i < 192; i += blockDim.x) sw2[i] = M.w2[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* const U = sU[wave]; float* const V = sV[wave];
    float acc0[16], acc1[16], acc2[3];
#pragma unroll
    for (int j = 0; j < 16; j++) { acc0[j] = 0.f; acc1[j] = 0.f; }
    acc2[0] = acc2[1] = acc2[2] = 0.f;
    const int tiles = (n + 63) / 64;
    for (int tile = blockIdx.x * MR_BW_WAVES + wave; tile < tiles; tile += gridDim.x * MR_BW_WAVES) {
        const int i = tile * 64 + lane;
        const bool live = i < n;
        float a0[32], h1[32], h2[32], gz2[6], gh2[32], gh1[32], ga0[32];
        float x[3] = {0.f, 0.f, 0.f}, xraw[3] = {0.f, 0.f, 0.f};
        // Round 6: a tile whose 64 points all carry a zero cotangent (the background pixels of a training view: 60 % of an image; the loss is taken over foreground
        // pixels) adds exact zeros to every gradient — skipped as a wave; its position gradient is the zero the full computation writes
        {
            bool nz = false;
            if (live) {
#pragma unroll
                for (int o = 0; o < 6; o++) nz = nz || gout[6 * (size_t)i + o] != 0.f;
            }
            if (!__ballot(nz)) {
                if (g_pos && live) { g_pos[3 * (size_t)i] = 0.f; g_pos[3 * (size_t)i + 1] = 0.f; g_pos[3 * (size_t)i + 2] = 0.f; }
                continue;
            }
        }
        if (live) {
#pragma unroll
            for (int d = 0; d < 3; d++) { xraw[d] = (pos[3 * (size_t)i + d] - M.aabb_min[d]) / (M.aabb_max[d] - M.aabb_min[d]); x[d] = fminf(fmaxf(xraw[d], 0.f), 1.f); }
        }
        // forward recompute
        for (int lv = 0; lv < MR_LEVELS; lv++) {
            const float scale = L.scale[lv]; const uint32_t res = L.res[lv], size = L.size[lv];
            const __half2* g = M.grid + L.offset[lv];
            float p[3]; uint32_t pg[3];
#pragma unroll
            for (int d = 0; d < 3; d++) { float q = fmaf(scale, x[d], 0.5f); float fl = floorf(q); pg[d] = (uint32_t)(int)fl; p[d] = q - fl; }
            uint32_t ci[8]; corner_indices(size, res, pg, ci);
            __half2 cv[8];
#pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) cv[idx] = g[ci[idx]];
            __half2 r = __floats2half2_rn(0.f, 0.f);
#pragma unroll
            for (uint32_t idx = 0; idx < 8; idx++) {
                float w = 1.f;
#pragma unroll
                for (int d = 0; d < 3; d++) w *= (idx & (1u << d)) == 0 ? 1 - p[d] : p[d];
                r = __hadd2(r, weighted_half2(w, cv[idx]));
            }
            a0[2 * lv] = __low2float(r); a0[2 * lv + 1] = __high2float(r);
        }
        float z2[6];
        for (int o = 0; o < 32; o++) { float acc = 0.f; for (int k = 0; k < 32; k++) acc = fmaf(a0[k], sw0[o * 32 + k], acc); h1[o] = fmaxf(acc, 0.f); }
        for (int o = 0; o < 32; o++) { float acc = 0.f; for (int k = 0; k < 32; k++) acc = fmaf(h1[k], sw1[o * 32 + k], acc); h2[o] = fmaxf(acc, 0.f); }
        for (int o = 0; o < 6; o++) { float acc = 0.f; for (int k = 0; k < 32; k++) acc = fmaf(h2[k], sw2[o * 32 + k], acc); z2[o] = acc; }
        // backward through the MLP (a dead lane carries zeros)
        for (int o = 0; o < 6; o++) { float sg = mrf_sigmoid(z2[o]); gz2[o] = live ? gout[6 * (size_t)i + o] * (M.mx[o] - M.mn[o]) * sg * (1.f - sg) : 0.f; }
        for (int k = 0; k < 32; k++) { float acc = 0.f; for (int o = 0; o < 6; o++) acc += gz2[o] * sw2[o * 32 + k]; gh2[k] = h2[k] > 0.f ? acc : 0.f; }
        for (int k = 0; k < 32; k++) { float acc = 0.f; for (int o = 0; o < 32; o++) acc += gh2[o] * sw1[o * 32 + k]; gh1[k] = h1[k] > 0.f ? acc : 0.f; }
        for (int k = 0; k < 32; k++) { float acc = 0.f; for (int o = 0; o < 32; o++) acc += gh1[o] * sw0[o * 32 + k]; ga0[k] = acc; }
        // Position gradient (the reference's sample() is differentiable in its argument: tcnn's HashGrid returns dL/dx, the x128 / /128 hooks of
        // render_helper.py:41,78-80 cancel on this route, and kd / ks losses reach `vertices_offsets` through dr.interpolate, nerf/renderer.py:1017-1018).
        // d enc[f] / d x_d = scale * sum_corners (+-1 along d) * (product of the other two weights) * table[corner][f]   (tcnn grid.h, dy_dx with
        // pos_derivative = 1 for linear interpolation; tcnn differences the fp16 entries and accumulates in fp16, here the sum is fp32);
        // d x / d pos = 1 / (aabb_max - aabb_min), torch.clamp passes the gradient on [0, 1] inclusive and blocks it outside.
        if (g_pos && live) {
            float gx[3] = {0.f, 0.f, 0.f};
            for (int lv = 0; lv < MR_LEVELS; lv++) {
                const float g0 = ga0[2 * lv], g1 = ga0[2 * lv + 1];
                if (g0 == 0.f && g1 == 0.f) continue;
                const float scale = L.scale[lv]; const uint32_t res = L.res[lv], size = L.size[lv];
                const __half2* g = M.grid + L.offset[lv];
                float p[3]; uint32_t pg[3];
#pragma unroll
                for (int d = 0; d < 3; d++) { float q = fmaf(scale, x[d], 0.5f); float fl = floorf(q); pg[d] = (uint32_t)(int)fl; p[d] = q - fl; }
                float lx[3] = {0.f, 0.f, 0.f};
                uint32_t ci[8]; corner_indices(size, res, pg, ci);
#pragma unroll
                for (uint32_t idx = 0; idx < 8; idx++) {
                    float wd[3];
#pragma unroll
                    for (int d = 0; d < 3; d++) wd[d] = (idx & (1u << d)) == 0 ? 1 - p[d] : p[d];
                    const __half2 v = g[ci[idx]];
                    const float sv = g0 * __low2float(v) + g1 * __high2float(v);
                    lx[0] += ((idx & 1u) ? sv : -sv) * (wd[1] * wd[2]);
                    lx[1] += ((idx & 2u) ? sv : -sv) * (wd[0] * wd[2]);
                    lx[2] += ((idx & 4u) ? sv : -sv) * (wd[0] * wd[1]);
                }
#pragma unroll
                for (int d = 0; d < 3; d++) gx[d] += scale * lx[d];
            }
#pragma unroll
            for (int d = 0; d < 3; d++)
                g_pos[3 * (size_t)i + d] = (xraw[d] >= 0.f && xraw[d] <= 1.f) ? gx[d] / (M.aabb_max[d] - M.aabb_min[d]) : 0.f;
        }
        // weight gradients: three staged outer-product sums over the wave's 64 points (wave-synchronous: a wave owns its U / V)
        {   // gW2[o][k] += gz2[o] * h2[k]   (6 x 32 = 192 entries: lane owns e = lane + 64 j, j < 3)
            for (int o = 0; o < 6; o++) U[lane * MR_BW_LD + o] = gz2[o];
            for (int k = 0; k < 32; k++) V[lane * MR_BW_LD + k] = h2[k];
            __builtin_amdgcn_wave_barrier();
            for (int p = 0; p < 64; p++) {
                const float vk = V[p * MR_BW_LD + (lane & 31)];
#pragma unroll
                for (int j = 0; j < 3; j++) acc2[j] = fmaf(U[p * MR_BW_LD + (lane >> 5) + 2 * j], vk, acc2[j]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        {   // gW1[o][k] += gh2[o] * h1[k]
            for (int o = 0; o < 32; o++) U[lane * MR_BW_LD + o] = gh2[o];
            for (int k = 0; k < 32; k++) V[lane * MR_BW_LD + k] = h1[k];
            __builtin_amdgcn_wave_barrier();
            for (int p = 0; p < 64; p++) {
                const float vk = V[p * MR_BW_LD + (lane & 31)];
#pragma unroll
                for (int j = 0; j < 16; j++) acc1[j] = fmaf(U[p * MR_BW_LD + (lane >> 5) + 2 * j], vk, acc1[j]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        {   // gW0[o][k] += gh1[o] * a0[k]
            for (int o = 0; o < 32; o++) U[lane * MR_BW_LD + o] = gh1[o];
            for (int k = 0; k < 32; k++) V[lane * MR_BW_LD + k] = a0[k];
            __builtin_amdgcn_wave_barrier();
            for (int p = 0; p < 64; p++) {
                const float vk = V[p * MR_BW_LD + (lane & 31)];
#pragma unroll
                for (int j = 0; j < 16; j++) acc0[j] = fmaf(U[p * MR_BW_LD + (lane >> 5) + 2 * j], vk, acc0[j]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // hash-grid gradient: fp32 atomics into the master-precision gradient table (tcnn accumulates its grid gradient the same way). The 64
        // points of a wave are neighbouring pixels: on the coarse levels they fall into a handful of cells, and 164 M scattered global atomics
        // (with that contention) were what the kernel waited for. Levels < MR_BW_COARSE therefore go through a per-wave LDS table (the wave's
        // staging area, free after the outer products: open addressing, two probes, overflow straight to global memory), flushed per tile.
        if (g_params) {
            // Round 6 (profiles/r06_atomic_rate.txt): the memory side takes 21 G scattered atomic REQUESTS per second, and the lanes of one instruction that fall into the
            // same 32-byte sector are one request. A table entry's two features are 8 consecutive bytes: every update is issued by a PAIR of lanes (even lane: feature 0,
            // odd lane: feature 1 of the same entry; first the even lane's entry, then the odd lane's) — half the requests for the same adds, in the same order per word.
            int* const keys = reinterpret_cast<int*>(U); float* const vals = V;
            for (int e = lane; e < MR_BW_TABLE; e += 64) { keys[e] = -1; vals[2 * e] = 0.f; vals[2 * e + 1] = 0.f; }
            __builtin_amdgcn_wave_barrier();
            const int odd = lane & 1;
            for (int lv = 0; lv < MR_LEVELS; lv++) {      // every lane walks every level (a dead or gradient-free lane contributes nothing): the pairs must stay together
                const float scale = L.scale[lv]; const uint32_t res = L.res[lv], size = L.size[lv];
                float p[3]; uint32_t pg[3];
#pragma unroll
                for (int d = 0; d < 3; d++) { float q = fmaf(scale, x[d], 0.5f); float fl = floorf(q); pg[d] = (uint32_t)(int)fl; p[d] = q - fl; }
                const float g0 = ga0[2 * lv], g1 = ga0[2 * lv + 1];
                const bool lvalid = live && !(g0 == 0.f && g1 == 0.f);
                if (!__ballot(lvalid)) continue;
                uint32_t ci[8]; corner_indices(size, res, pg, ci);
#pragma unroll
                for (uint32_t idx = 0; idx < 8; idx++) {
                    float w = 1.f;
#pragma unroll
                    for (int d = 0; d < 3; d++) w *= (idx & (1u << d)) == 0 ? 1 - p[d] : p[d];
                    const uint32_t e = L.offset[lv] + ci[idx];
                    bool direct = lvalid;
                    if (lv < MR_BW_COARSE && lvalid) {
                        uint32_t h = (e * 2654435761u) >> (32 - MR_BW_TABLE_LOG2);
#pragma unroll
                        for (int probe = 0; probe < 2 && direct; probe++) {
                            const int old = atomicCAS(&keys[h], -1, (int)e);
                            if (old == -1 || old == (int)e) { atomicAdd(&vals[2 * h], w * g0); atomicAdd(&vals[2 * h + 1], w * g1); direct = false; }
                            h = (h + 1) & (MR_BW_TABLE - 1);
                        }
                    }
                    const float v0 = w * g0, v1 = w * g1;
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const int src = (lane & ~1) | r;
                        const uint32_t e_s = (uint32_t)__shfl((int)e, src, 64);
                        const int d_s = __shfl(direct ? 1 : 0, src, 64);
                        const float a_s = __shfl(v0, src, 64), b_s = __shfl(v1, src, 64);
                        if (d_s) atomicAdd(&g_params[2 * (size_t)e_s + odd], odd ? b_s : a_s);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int j = lane; j < 2 * MR_BW_TABLE; j += 64) {      // lane pairs again: entry j / 2, feature j % 2
                const int key = keys[j >> 1];
                if (key >= 0) atomicAdd(&g_params[2 * (size_t)key + (j & 1)], vals[j]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // reduce the four waves' register accumulators through LDS (entry e = lane + 64 j  <->  [o = e / 32][k = e % 32]), one global atomic per entry
    __syncthreads();
    float* red = &sU[0][0];            // 2 240 floats needed, 4 * 64 * 33 available
    for (int i = threadIdx.x; i < 2240; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) { atomicAdd(&red[lane + 64 * j], acc0[j]); atomicAdd(&red[1024 + lane + 64 * j], acc1[j]); }
#pragma unroll
    for (int j = 0; j < 3; j++) atomicAdd(&red[2048 + lane + 64 * j], acc2[j]);
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += blockDim.x) { if (g_w0 && red[k] != 0.f) atomicAdd(&g_w0[k], red[k]); if (g_w1 && red[1024 + k] != 0.f) atomicAdd(&g_w1[k], red[1024 + k]); }
    for (int k = threadIdx.x; k < 192; k += blockDim.x) if (g_w2 && red[2048 + k] != 0.f) atomicAdd(&g_w2[k], red[2048 + k]);
}

}  // namespace mr

using namespace mr;

extern "C" {

int mirres_final_shading_bwd(mirres_ctx_t* ctx, const float* occ, const float* normal, const float* ray_dir, const float* kd,
                             const float* rough_metal, const float* final_dir, const float* final_dist, const float* final_Li,
                             const float* g_color, const float* g_diff, const float* g_spec, float* g_normal, float* g_kd,
                             float* g_rough_metal, float* g_final_Li, void* stream) {
    if (!ctx || !occ || !normal || !ray_dir || !kd || !rough_metal || !final_dir || !final_dist || !final_Li || !g_color || !g_diff || !g_spec) {
        set_error("mirres_final_shading_bwd: null"); return MIRRES_E_ARG;
    }
    const int N = (int)ctx->N;
    k_final_shading_bwd<<<grid_for(N, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(N, occ, normal, ray_dir, kd, rough_metal, final_dir, final_dist, final_Li, g_color,
                                                                                     g_diff, g_spec, g_normal, g_kd, g_rough_metal, g_final_Li);
    MR_LAUNCH_CHECK("final_shading_bwd");
    return MIRRES_OK;
}

int mirres_matnet_bwd(const mirres_matnet_t* m, const float* pos, int n, const float* grad_out, float* g_params_f32, float* g_w0,
                      float* g_w1, float* g_w2, float* g_pos, void* stream) {
    if (!m || !pos || !grad_out || n < 0) { set_error("mirres_matnet_bwd: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    MatNetB M; M.grid = reinterpret_cast<const __half2*>(m->grid_f16); M.w0 = m->w0; M.w1 = m->w1; M.w2 = m->w2;
    for (int i = 0; i < 3; i++) { M.aabb_min[i] = m->aabb_min[i]; M.aabb_max[i] = m->aabb_max[i]; }
    for (int i = 0; i < 6; i++) { M.mn[i] = m->out_min[i]; M.mx[i] = m->out_max[i]; }
    int grd = grid_for(n, MR_BLOCK); if (grd > 256 * 4) grd = 256 * 4;      // workgroups loop over point tiles and keep the weight gradients in registers
    k_matnet_bwd<<<grd, MR_BLOCK, 0, (hipStream_t)stream>>>(M, host_levels_b(), pos, n, grad_out, g_params_f32, g_w0, g_w1, g_w2, g_pos);
    MR_LAUNCH_CHECK("matnet_bwd");
    return MIRRES_OK;
}

}  // extern "C"
