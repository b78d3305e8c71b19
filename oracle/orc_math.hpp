// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the reference's hot-path arithmetic (brabbitdousha/MIRReS-ReSTIR_Nerf_mesh).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// PARITY UNPINNED: the reference holds no tests / golden vectors for this path and its Slang->CUDA
// kernels cannot be compiled or imported here; the oracle is pinned only by
// known-answer values derived from the reference formulas and by independent invariants (tests/).
// What IS pinned by reference code run in the build container (tests/golden/gen_*.py): the frame loop and its pre / post
// processing (orc_render against the reference's own Python loop executed over these kernels), the a-trous driver, the
// importance-table construction, and nerf/render_dump.py (orc_dump.hpp).
//
// Fixed floating-point policy (shared, by contract, with the HIP product — not by shared code):
//   * fp32 everywhere, IEEE division and sqrt, no FMA contraction (-ffp-contract=off),
//     no fast-math, denormals kept;
//   * dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   * cross(a,b)    = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
//   * normalize(v)  = v * (1.0f / sqrtf(dot(v,v)))
//   * lerp(a,b,t)   = a + (b - a) * t ;  saturate(x) = fminf(fmaxf(x,0),1)
//   * reflect(i,n)  = i - n * (2.0f * dot(n,i))
//   * min/max       = fminf/fmaxf
// The reference's nvcc flags are unknown, so "bit-exact" is defined against
// this policy.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
// acos / atan2 / sin / cos / exp / exp2 / integer powers: the SAME fixed arithmetic as the HIP product (include/mirres_fmath.h, <= 2 ulp from
// the correctly rounded functions — oracle/fmath_check.cpp), so that discrete sampler decisions can be compared pixel by pixel.
#include "../include/mirres_fmath.h"

namespace orc {

struct f2 { float x, y; };
struct f3 { float x, y, z; };

static inline f3 mk3(float a, float b, float c) { f3 r = {a, b, c}; return r; }
static inline f3 mk3(float a) { f3 r = {a, a, a}; return r; }
static inline f2 mk2(float a, float b) { f2 r = {a, b}; return r; }

static inline f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
static inline f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
static inline f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
static inline f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
static inline f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
static inline f3& operator*=(f3& a, float s) { a = a * s; return a; }

static inline float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
static inline f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 normalize(f3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
static inline float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
static inline float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
static inline int clampi(int x, int a, int b) { return x < a ? a : (x > b ? b : x); }
static inline float lerpf(float a, float b, float t) { return a + (b - a) * t; }
static inline f3 reflect(f3 i, f3 n) { return i - n * (2.0f * dot(n, i)); }
static inline bool is_black(f3 v) { return !(v.x != 0.f) && !(v.y != 0.f) && !(v.z != 0.f); }  // helperDi.slang:397-405

// utils/helperDi.slang:104-107
static inline float luminance(f3 v) { return v.x * 0.212671f + v.y * 0.715160f + v.z * 0.072169f; }

// ---------------------------------------------------------------- RNG  (utils/random.slang:2-74)
static inline uint32_t interleave_32bit(uint32_t vx, uint32_t vy) {
    uint32_t x = vx & 0x0000ffffu, y = vy & 0x0000ffffu;
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    y = (y | (y << 8)) & 0x00FF00FFu; y = (y | (y << 4)) & 0x0F0F0F0Fu;
    y = (y | (y << 2)) & 0x33333333u; y = (y | (y << 1)) & 0x55555555u;
    return x | (y << 1);
}
static inline uint32_t seed_generator(uint32_t px, uint32_t py, uint32_t sample) {
    uint32_t v0 = interleave_32bit(px, py), v1 = sample, sum = 0;
    const uint32_t k0 = 0xa341316cu, k1 = 0xc8013ea4u, k2 = 0xad90777du, k3 = 0x7e95761eu;
    for (int i = 0; i < 16; i++) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + k0) ^ (v1 + sum) ^ ((v1 >> 5) + k1);
        v1 += ((v0 << 4) + k2) ^ (v0 + sum) ^ ((v0 >> 5) + k3);
    }
    return v0;
}
static inline float next1d(uint32_t& s) {
    s = 1664525u * s + 1013904223u;
    return (float)(s >> 8) * 0x1p-24f;
}

// ---------------------------------------------------------------- octahedral (helperDi.slang:109-134)
static inline f2 oct_encode(f3 n) {
    float l1 = (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z);
    n = mk3(n.x / l1, n.y / l1, n.z / l1);
    // oct_wrap(n.xy) = (1 - abs(n.yx)) * sign(n.xy), sign(0) = +1
    float wx = (1.0f - fabsf(n.y)) * (n.x >= 0.0f ? 1.0f : -1.0f);
    float wy = (1.0f - fabsf(n.x)) * (n.y >= 0.0f ? 1.0f : -1.0f);
    float ex = n.z >= 0.0f ? n.x : wx;
    float ey = n.z >= 0.0f ? n.y : wy;
    return mk2(ex * 0.5f + 0.5f, ey * 0.5f + 0.5f);
}
static inline f3 oct_decode(f2 f) {
    float fx = f.x * 2.0f - 1.0f, fy = f.y * 2.0f - 1.0f;
    f3 n = mk3(fx, fy, (1.0f - fabsf(fx)) - fabsf(fy));
    float t = clampf(-n.z, 0.0f, 1.0f);
    n.x += (n.x >= 0.0f ? -t : t);
    n.y += (n.y >= 0.0f ? -t : t);
    return normalize(n);
}
// lightDi.slang:432-436
static inline f3 ngp_dir(f3 d) { return mk3(-d.x, d.z, d.y); }

// ---------------------------------------------------------------- fp16 (software, RNE) for the tcnn hash grid
static inline uint16_t f64_to_f16(double d) {
    // round-to-nearest-even double -> IEEE binary16 (handles subnormals, overflow->inf)
    uint64_t b; std::memcpy(&b, &d, 8);
    uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
    int64_t e = (int64_t)((b >> 52) & 0x7ff);
    uint64_t m = b & 0xfffffffffffffULL;
    if (e == 0x7ff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
    if (e == 0 && m == 0) return sign;
    int64_t E = e - 1023;             // unbiased
    uint64_t sig = m | (1ULL << 52);  // 53-bit significand (denormal doubles are far below half range)
    if (e == 0) return sign;          // double subnormal -> 0 in half
    if (E > 15) return (uint16_t)(sign | 0x7c00u);
    int shift;                        // bits to drop from the 53-bit significand
    int64_t he;                       // half biased exponent
    if (E >= -14) { shift = 42; he = E + 15; }
    else { shift = (int)(42 + (-14 - E)); he = 0; if (shift > 63) return sign; }
    uint64_t q = sig >> shift, rem = sig & ((1ULL << shift) - 1), half = 1ULL << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q++;
    uint32_t h;
    if (he > 0) { h = (uint32_t)((he << 10) + (q - 1024)); }  // q in [1024,2048]; carry handled by addition
    else h = (uint32_t)q;                                     // subnormal (q may reach 1024 -> smallest normal)
    if (h >= 0x7c00u) h = 0x7c00u;
    return (uint16_t)(sign | h);
}
static inline float f16_to_f32(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, r;
    if (e == 0) {
        if (m == 0) r = s;
        else { int k = 0; while (!(m & 0x400u)) { m <<= 1; k++; } m &= 0x3ffu; r = s | ((uint32_t)(127 - 15 - k + 1) << 23) | (m << 13); }
    } else if (e == 31) r = s | 0x7f800000u | (m << 13);
    else r = s | ((e + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &r, 4); return f;
}
static inline uint16_t f32_to_f16(float f) { return f64_to_f16((double)f); }
static inline uint16_t f16_add(uint16_t a, uint16_t b) { return f64_to_f16((double)f16_to_f32(a) + (double)f16_to_f32(b)); }

}  // namespace orc
