// device_math.hpp — gfx950 device-side scalar math for the ReSTIR path tracer.
//
// Floating-point contract (DESIGN.md §FP policy): fp32, IEEE division/sqrt (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), FMA contraction OFF for the whole library (-ffp-contract=off) so
// that traversal decisions are reproducible; explicit fmaf only where the algorithm asks for it (hash-grid
// position, MLP accumulation = MFMA semantics). Formulas cite the reference Slang they implement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MR_DEV __device__ __forceinline__

namespace mr {

// ---- division and square root. MR_LEAN_FP (set by the shading translation units before this header; the traversal keeps the compiler's
// operations) selects short sequences that return the SAME bits as the compiler's IEEE-754 operations for every operand the renderer produces:
//   mr_div:  v_rcp_f32, one Newton step on the reciprocal, q = a * r, one residual correction, v_div_fixup_f32 — 7 instructions / ~30 issue cycles
//            against the compiler's 11 / ~46 (two v_div_scale, a second correction, v_div_fmas). Proved on the hardware by exhaustion
//            (scripts/ubench/div_exhaustive.hip): all 2^46 pairs of significands agree with `a / b`, i.e. every pair of normal operands with a normal
//            quotient while no intermediate leaves the normal range — operands and quotient within 2^-102 .. 2^102; zeros, infinities and NaNs by
//            v_div_fixup_f32 exactly as in the compiler's sequence. Beyond 2^+-102 (nothing in a frame: radiances, pdfs, cosines) it can be off:
//            v_rcp_f32 flushes a denormal divisor to zero, so a / denormal comes out as what a / 0 gives (inf, or NaN for 0 / denormal) where IEEE division
//            returns a large finite number. Where such a quotient could reach a reservoir it is harmless by construction of the callers: a NaN or inf
//            weight zeroes the reservoir (store_ris, as InitialResampling.slang:285-293 does), and a pdf below 1.2e-38 belongs to a sample that is
//            never selected. The traversal kernels, whose hostile-geometry tests do feed denormal reciprocals, keep the compiler's division.
//   mr_sqrt: v_rsq_f32, s = x * y, one residual correction with y / 2; +-0 and +inf passed through — 7 instructions / ~28 cycles against 17 / ~60.
//            All 2^24 significand / exponent-parity cases agree with sqrtf; a denormal argument is returned as it is (sqrtf would give ~1e-20).
#ifndef MR_LEAN_FP
#define MR_LEAN_FP 0
#endif
// the short reciprocal under its own name, for single call sites of translation units that otherwise keep the compiler's operations
MR_DEV float lean_rcp(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r), b, 1.0f);
}
#if MR_LEAN_FP
MR_DEV float mr_div(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    const float q = a * r;
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-b, q, a), r, q), b, a);
}
MR_DEV float mr_rcp(float b) {   // 1 / b: q = 1 * r exactly
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r), b, 1.0f);
}
MR_DEV float mr_sqrt(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    s = __builtin_fmaf(__builtin_fmaf(-s, s, x), 0.5f * y, s);
    return __builtin_amdgcn_classf(x, 0x2f0) ? x : s;      // +-0, +-denormal, +inf
}
#else
MR_DEV float mr_div(float a, float b) { return a / b; }
MR_DEV float mr_rcp(float b) { return 1.0f / b; }
MR_DEV float mr_sqrt(float x) { return sqrtf(x); }
#endif
}  // namespace mr

// The transcendental functions of the path (acos, atan2, sin, cos, exp, exp2, integer powers): fixed sequences of IEEE operations shared with the
// host-side checker of the tests — include/mirres_fmath.h.  Its square root (acos: argument in [2^-25, 1/2] or zero) may be the short sequence above: same bits there.
#define MRF_SQRT(x) ::mr::mr_sqrt(x)
#include "mirres_fmath.h"

namespace mr {

struct v2 { float x, y; };
struct v3 { float x, y, z; };

MR_DEV v3 V3(float a, float b, float c) { v3 r; r.x = a; r.y = b; r.z = c; return r; }
MR_DEV v3 V3(float a) { return V3(a, a, a); }
MR_DEV v2 V2(float a, float b) { v2 r; r.x = a; r.y = b; return r; }
MR_DEV v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
MR_DEV v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
MR_DEV v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
MR_DEV v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
MR_DEV v3 operator*(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
MR_DEV v3 operator/(v3 a, float s) { return V3(mr_div(a.x, s), mr_div(a.y, s), mr_div(a.z, s)); }
MR_DEV v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
MR_DEV float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MR_DEV float dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
MR_DEV v3 cross(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MR_DEV v3 normalize(v3 v) { float inv = mr_rcp(mr_sqrt(dot(v, v))); return v * inv; }
MR_DEV float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
MR_DEV float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
MR_DEV int clampi(int x, int a, int b) { return x < a ? a : (x > b ? b : x); }
MR_DEV float lerpf(float a, float b, float t) { return a + (b - a) * t; }
MR_DEV v3 reflect(v3 i, v3 n) { return i - n * (2.0f * dot(n, i)); }
MR_DEV bool is_black(v3 v) { return !(v.x != 0.f) && !(v.y != 0.f) && !(v.z != 0.f); }
MR_DEV float luminance(v3 v) { return v.x * 0.212671f + v.y * 0.715160f + v.z * 0.072169f; }  // helperDi.slang:104-107

MR_DEV v3 ld3(const float* p, size_t i) { return V3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
MR_DEV void st3(float* p, size_t i, v3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

// ---- RNG: TEA-16 seed + LCG stream (utils/random.slang:2-74)
MR_DEV uint32_t interleave16(uint32_t vx, uint32_t vy) {
    uint32_t x = vx & 0xffffu, y = vy & 0xffffu;
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    y = (y | (y << 8)) & 0x00FF00FFu; y = (y | (y << 4)) & 0x0F0F0F0Fu; y = (y | (y << 2)) & 0x33333333u; y = (y | (y << 1)) & 0x55555555u;
    return x | (y << 1);
}
MR_DEV uint32_t seed_generator(uint32_t px, uint32_t py, uint32_t sampleNumber) {
    uint32_t v0 = interleave16(px, py), v1 = sampleNumber, sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
MR_DEV float rnd(uint32_t& s) { s = 1664525u * s + 1013904223u; return (float)(s >> 8) * 0x1p-24f; }

// ---- octahedral direction coding (helperDi.slang:109-134)
MR_DEV v2 oct_encode(v3 n) {
    float l1 = (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z);
    float nx = mr_div(n.x, l1), ny = mr_div(n.y, l1), nz = mr_div(n.z, l1);
    float wx = (1.0f - fabsf(ny)) * (nx >= 0.0f ? 1.0f : -1.0f);
    float wy = (1.0f - fabsf(nx)) * (ny >= 0.0f ? 1.0f : -1.0f);
    float ex = nz >= 0.0f ? nx : wx, ey = nz >= 0.0f ? ny : wy;
    return V2(ex * 0.5f + 0.5f, ey * 0.5f + 0.5f);
}
MR_DEV v3 oct_decode(v2 f) {
    float fx = f.x * 2.0f - 1.0f, fy = f.y * 2.0f - 1.0f;
    v3 n = V3(fx, fy, (1.0f - fabsf(fx)) - fabsf(fy));
    float t = clampf(-n.z, 0.0f, 1.0f);
    n.x += (n.x >= 0.0f ? -t : t);
    n.y += (n.y >= 0.0f ? -t : t);
    return normalize(n);
}
MR_DEV v3 ngp_dir(v3 d) { return V3(-d.x, d.z, d.y); }  // lightDi.slang:432-436

// ---- wave64 helpers
MR_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// compacted append: every lane with `want` gets a unique slot in [*counter, ...); one atomic per wave.
MR_DEV uint32_t wave_append(uint32_t* counter, bool want, uint32_t n = 1) {
    // n may differ per lane (exclusive prefix over the wave via shuffles)
    uint32_t mine = want ? n : 0u;
    uint32_t incl = mine;
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    uint32_t total = __shfl(incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(counter, total);
    base = __shfl(base, 63, 64);
    return base + incl - mine;
}

// Block-level compacted append: ONE global atomic per workgroup (a single queue-head word saturates at ~88 atomics/us on MI355X, and the
// per-wave version made every ray-generating kernel atomic-bound: 17 k wave atomics = 195 us for a kernel that moves 150 MB).
// Every thread of the block must call it (two __syncthreads). `n` rays are reserved for lanes with `want`.
MR_DEV uint32_t block_append(uint32_t* counter, bool want, uint32_t n = 1) {
    __shared__ uint32_t s_wave[17];
    uint32_t mine = want ? n : 0u;
    uint32_t incl = mine;
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < nwaves; w++) { uint32_t t = s_wave[w]; s_wave[w] = tot; tot += t; }
        s_wave[16] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    const uint32_t r = s_wave[16] + s_wave[wave] + incl - mine;
    __syncthreads();   // s_wave may be reused by a second append in the same kernel
    return r;
}

}  // namespace mr
