// ORACLE — TEST INFRASTRUCTURE ONLY (tests/test_fmath.py, tests/test_gpu_fmath.py).
// Two jobs for include/mirres_fmath.h (the fixed transcendental arithmetic shared by the HIP product and the oracle):
//   (1) accuracy: maximum error in ulps against double-precision libm (glibc, < 1 double ulp, i.e. exact for this purpose) over each function's
//       whole domain (exhaustive over binary32 for the one-argument functions, dense structured + random sweeps for atan2);
//   (2) host-side checksums over ranges of argument bit patterns, compared by the GPU test with mirres_fmath_checksum (same arithmetic on gfx950).
// Build: g++ -O2 -ffp-contract=off -mfma -fopenmp -shared -fPIC (oracle/Makefile -> libfmathcheck.so).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <algorithm>
#include "../include/mirres_fmath.h"

namespace {

inline float asf(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t asu(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline uint32_t canon(float f) { return f != f ? 0x7fc00000u : asu(f); }

// error of `got` against the exact value `want`, in units of the binary32 ulp at `want` (subnormal spacing below 2^-126)
inline double ulp_err(float got, double want) {
    if (want != want) return got != got ? 0.0 : 1e30;
    if (std::isinf(want)) return (std::isinf(got) && (got > 0) == (want > 0)) ? 0.0 : 1e30;
    if (got != got || std::isinf(got)) {
        if (std::isinf(got) && std::fabs(want) > 3.4028234663852886e38) return 0.0;   // rounds to infinity in binary32
        return 1e30;
    }
    int e; std::frexp(want, &e);            // |want| = m 2^e, m in [1/2, 1)
    int ue = std::max(e - 1, -126) - 23;
    return std::fabs((double)got - want) / std::ldexp(1.0, ue);
}

// one-argument functions by id (shared with csrc/selfcheck.hip: mirres_fmath_checksum / mirres_fmath_eval)
inline float eval1(int fn, float x) {
    switch (fn) {
        case 0: return mrf_sin(x);
        case 1: return mrf_cos(x);
        case 2: return mrf_acos(x);
        case 3: return mrf_exp(x);
        case 4: return mrf_exp2(x);
        case 5: return mrf_pow5(x);
        case 6: return mrf_pow2k(x, 3);
        case 7: return mrf_pow2k(x, 7);
        case 8: return mrf_sigmoid(x);
        default: return 0.f;
    }
}
inline double exact1(int fn, double x) {
    switch (fn) {
        case 0: return std::sin(x);
        case 1: return std::cos(x);
        case 2: return std::acos(x);
        case 3: return std::exp(x);
        case 4: return std::exp2(x);
        case 5: return x * x * x * x * x;
        case 6: return std::pow(x, 8.0);
        case 7: return std::pow(x, 128.0);
        case 8: return 1.0 / (1.0 + std::exp(-x));
        default: return 0.0;
    }
}
// the second argument of the two-argument sweep is derived from the index so that host and device enumerate the same pairs
inline uint32_t pair_hash(uint32_t i) { i ^= i >> 16; i *= 0x7feb352du; i ^= i >> 15; i *= 0x846ca68bu; i ^= i >> 16; return i; }

}  // namespace

extern "C" {

// max ulp error of function fn over the bit patterns [first, first + count) (both signs are separate ranges); *worst_bits = argument of the maximum
double fmath_max_ulp(int fn, uint32_t first, uint64_t count, uint32_t* worst_bits) {
    double worst = 0.0; uint32_t wb = first;
#pragma omp parallel
    {
        double w = 0.0; uint32_t b = first;
#pragma omp for schedule(static) nowait
        for (int64_t i = 0; i < (int64_t)count; i++) {
            const uint32_t u = first + (uint32_t)i;
            const float x = asf(u);
            const double e = ulp_err(eval1(fn, x), exact1(fn, (double)x));
            if (e > w) { w = e; b = u; }
        }
#pragma omp critical
        if (w > worst) { worst = w; wb = b; }
    }
    if (worst_bits) *worst_bits = wb;
    return worst;
}

// atan2 accuracy: (mode 0) y = all bit patterns in [first, first+count), x = 1 (the atan core incl. the octant fold through y > x);
// (mode 1) pseudo-random pairs (y, x) from the index: both signs, magnitudes 2^-20 .. 2^20; (mode 2) unit vectors (cos a, sin a) perturbed to floats
double fmath_atan2_max_ulp(int mode, uint32_t first, uint64_t count, uint32_t* worst_y, uint32_t* worst_x) {
    double worst = 0.0; uint32_t wy = 0, wx = 0;
#pragma omp parallel
    {
        double w = 0.0; uint32_t by = 0, bx = 0;
#pragma omp for schedule(static) nowait
        for (int64_t i = 0; i < (int64_t)count; i++) {
            const uint32_t u = first + (uint32_t)i;
            float y, x;
            if (mode == 0) { y = asf(u); x = 1.0f; }
            else if (mode == 1) {
                const uint32_t a = pair_hash(u), b = pair_hash(u ^ 0x9e3779b9u);
                y = asf((a & 0x807fffffu) | ((107u + (a >> 23) % 41u) << 23));
                x = asf((b & 0x807fffffu) | ((107u + (b >> 23) % 41u) << 23));
            } else {
                const double ang = (double)u * (6.283185307179586 / 4294967296.0) - 3.141592653589793;
                y = (float)std::sin(ang); x = (float)std::cos(ang);
            }
            const double e = ulp_err(mrf_atan2(y, x), std::atan2((double)y, (double)x));
            if (e > w) { w = e; by = asu(y); bx = asu(x); }
        }
#pragma omp critical
        if (w > worst) { worst = w; wy = by; wx = bx; }
    }
    if (worst_y) *worst_y = wy;
    if (worst_x) *worst_x = wx;
    return worst;
}

// order-free checksum of (argument bits, result bits) over [first, first + count): sum of result * (2 * argument + 1) mod 2^64 (NaNs canonical).
// fn < 16: one-argument function ids above; fn = 16: atan2(y, x) with y = bits, x = float of pair_hash(bits); fn = 17: x / y with |x| in 2^-20 .. 2^20
// (IEEE division; the device side runs its short sequence mr_div); fn = 18: sqrtf.
unsigned long long fmath_checksum(int fn, uint32_t first, uint64_t count) {
    unsigned long long sum = 0;
#pragma omp parallel for schedule(static) reduction(+ : sum)
    for (int64_t i = 0; i < (int64_t)count; i++) {
        const uint32_t u = first + (uint32_t)i;
        float r;
        if (fn < 16) r = eval1(fn, asf(u));
        else if (fn == 16) r = mrf_atan2(asf(u), asf(pair_hash(u)));
        else if (fn == 17) { const uint32_t h = pair_hash(u); r = asf((h & 0x807fffffu) | ((107u + (h >> 23) % 41u) << 23)) / asf(u); }
        else r = sqrtf(asf(u));
        sum += (unsigned long long)canon(r) * (2ull * u + 1ull);
    }
    return sum;
}

void fmath_eval(int fn, const float* a, const float* b, float* out, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) out[i] = fn < 16 ? eval1(fn, a[i]) : (fn == 16 ? mrf_atan2(a[i], b[i]) : (fn == 17 ? b[i] / a[i] : sqrtf(a[i])));
}

}  // extern "C"
