/* mirres_fmath.h — the transcendental functions of the path as FIXED arithmetic: the same sequence of IEEE-754 binary32 operations
 * (+, -, *, /, sqrt, fma, round-to-nearest-integer, integer bit operations) on the gfx950 device and on the host, so that both return the same bits.
 *
 * Why it exists: the reference evaluates acos / atan2 / sin / cos (environment look-up and sampling, utils/lightDi.slang:119-132, 181-209, 312-330;
 * utils/brdf.slang:76-124), exp (EAWDenoise.slang:50-302; the material field's sigmoid, nerf/render_helper.py:93-117) and pow (res.slang:53-61 mFactor,
 * brdf*.slang evalFresnelSchlick, renderutils bilateral weights) through its CUDA math library.  Any two libraries (CUDA's, ocml, glibc) differ in the last
 * bits of these, and one ulp can flip a discrete choice of the sampler (a CDF bin, a reservoir pick).  The HIP product (csrc/device_*.hpp) and the CPU
 * oracle (oracle/orc_*.hpp) both include THIS header, compiled with -ffp-contract=off (every fusion below is an explicit fma), so the parity tests can
 * ask for equal decisions in every pixel.  Each function is within 2 ulp of the correctly rounded result on its stated domain (exhaustive / dense
 * sweeps against double-precision libm: oracle/fmath_check.cpp, tests/test_fmath.py), i.e. as close to the mathematical function as the libraries are.
 * Device and host bits are compared exhaustively on the hardware (mirres_fmath_checksum, tests/test_gpu_fmath.py).
 *
 * Not test infrastructure and not oracle code: a plain header of arithmetic, part of the product's interface directory.
 * Division and square root are the IEEE operations (`/`, sqrtf) unless the includer defines MRF_DIV / MRF_SQRT to sequences proved to return the same
 * bits on the operand ranges used here (csrc/device_math.hpp does, for the square root of acos: its argument lies in [2^-25, 1/2] or is zero).
 */
#ifndef MIRRES_FMATH_H
#define MIRRES_FMATH_H
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define MRF_FN __host__ __device__ __forceinline__
#else
#define MRF_FN static inline
#endif
#ifndef MRF_DIV
#define MRF_DIV(a, b) ((a) / (b))
#endif
#ifndef MRF_SQRT
#define MRF_SQRT(x) sqrtf(x)
#endif

MRF_FN float mrf_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MRF_FN uint32_t mrf_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
MRF_FN float mrf_float(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
MRF_FN float mrf_nan(void) { return mrf_float(0x7fc00000u); }
MRF_FN float mrf_inf(void) { return mrf_float(0x7f800000u); }
MRF_FN float mrf_abs(float x) { return mrf_float(mrf_bits(x) & 0x7fffffffu); }
MRF_FN float mrf_copysign(float mag, float sgn) { return mrf_float((mrf_bits(mag) & 0x7fffffffu) | (mrf_bits(sgn) & 0x80000000u)); }

/* ---------------------------------------------------------------------------------------------------------------- sin, cos
 * Domain |x| <= 8192 (the path's arguments are angles in [-2 pi, 2 pi]); beyond it, and for inf / NaN, NaN.
 * Reduction: n = rint(x * 2/pi), r = x - n * pi/2 with pi/2 in three parts (72 bits) and fma, |r| <= pi/4 + 2^-17.
 * Kernels: odd / even polynomials in r on that interval. */
MRF_FN float mrf_sin_kernel(float r) {
    const float s = r * r;
    float p = mrf_fma(s, -1.9515295891e-4f, 8.3321608736e-3f);
    p = mrf_fma(p, s, -1.6666654611e-1f);
    return mrf_fma(r * s, p, r);
}
MRF_FN float mrf_cos_kernel(float r) {
    const float s = r * r;
    float p = mrf_fma(s, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = mrf_fma(p, s, 4.166664568298827e-2f);
    return mrf_fma(s * s, p, mrf_fma(s, -0.5f, 1.0f));
}
MRF_FN void mrf_sincos(float x, float* sn, float* cs) {
    if (!(mrf_abs(x) <= 8192.0f)) { *sn = mrf_nan(); *cs = mrf_nan(); return; }
    if (x == 0.0f) { *sn = x; *cs = 1.0f; return; }                /* sin(-0) = -0 */
    const float fn = rintf(x * 0.636619772367581343f);
    float r = mrf_fma(fn, -1.57079637050628662109375f, x);
    r = mrf_fma(fn, 4.37113882867379090888e-8f, r);
    r = mrf_fma(fn, 1.71512451000588187e-15f, r);
    const int q = (int)fn;
    const float s = mrf_sin_kernel(r), c = mrf_cos_kernel(r);
    const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;
    *sn = (q & 2) ? -a : a;
    *cs = ((q + 1) & 2) ? -b : b;
}
MRF_FN float mrf_sin(float x) { float s, c; mrf_sincos(x, &s, &c); return s; }
MRF_FN float mrf_cos(float x) { float s, c; mrf_sincos(x, &s, &c); return c; }

/* ---------------------------------------------------------------------------------------------------------------- acos
 * asin(t) = t + t z P(z), z = t^2 <= 1/4.  |x| <= 1/2: pi/2 - asin(x); x > 1/2: 2 asin(sqrt((1 - x) / 2)); x < -1/2: pi - 2 asin(sqrt((1 + x) / 2)).
 * |x| > 1 and NaN: NaN (square root of a negative number), as libm. */
MRF_FN float mrf_asin_poly(float z) {
    float p = mrf_fma(z, 4.2163199048e-2f, 2.4181311049e-2f);
    p = mrf_fma(p, z, 4.5470025998e-2f);
    p = mrf_fma(p, z, 7.4953002686e-2f);
    p = mrf_fma(p, z, 1.6666752422e-1f);
    return p * z;
}
MRF_FN float mrf_acos(float x) {
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113882867379090888e-8f;
    const float ax = mrf_abs(x);
    if (ax <= 0.5f) {
        const float z = x * x;
        const float w = mrf_asin_poly(z);                       /* asin(x) = x + x w */
        return pio2_hi - (x - (pio2_lo - x * w));
    }
    const float z = (1.0f - ax) * 0.5f;                         /* exact for ax in [1/2, 1] */
    const float s = MRF_SQRT(z);
    const float w = mrf_asin_poly(z);
    if (x > 0.0f) return 2.0f * mrf_fma(s, w, s);
    if (x < 0.0f) return 3.14159274101257324219f - 2.0f * (s + mrf_fma(s, w, -pio2_lo));
    return mrf_nan();                                           /* NaN argument */
}

/* ---------------------------------------------------------------------------------------------------------------- atan2
 * One division: a = min(|x|,|y|) / max(|x|,|y|) in [0, 1]; atan(a) = a + a w Q(w), w = a^2, Q of degree 8 (weighted minimax fit, approximation
 * error 0.05 ulp); octant / quadrant by pi/2 - t, pi - t with the constants in two parts; sign of y.  atan2(+-0, +x) = +-0, atan2(+-0, -x) = +-pi,
 * atan2(0, 0) follows the same rules (no division by zero), both infinite: pi/4 folded likewise, NaN in -> NaN. */
MRF_FN float mrf_atan2(float y, float x) {
    if (x != x || y != y) return mrf_nan();
    const float ax = mrf_abs(x), ay = mrf_abs(y);
    const float M = ax > ay ? ax : ay, m = ax > ay ? ay : ax;
    float a;
    if (M == 0.0f) a = 0.0f;
    else if (m == mrf_inf()) a = 1.0f;
    else a = MRF_DIV(m, M);
    const float w = a * a;
    float p = mrf_fma(w, -0.001793615985661745f, 0.010914580896496773f);
    p = mrf_fma(p, w, -0.031177803874015808f);
    p = mrf_fma(p, w, 0.05795755609869957f);
    p = mrf_fma(p, w, -0.08403448015451431f);
    p = mrf_fma(p, w, 0.10952185094356537f);
    p = mrf_fma(p, w, -0.14264240860939026f);
    p = mrf_fma(p, w, 0.19998548924922943f);
    p = mrf_fma(p, w, -0.33333298563957214f);
    float t = mrf_fma(a * w, p, a);
    if (ay > ax) t = 1.57079637050628662109375f - (t + 4.37113882867379090888e-8f);
    if (mrf_bits(x) & 0x80000000u) t = 3.14159274101257324219f - (t + 8.74227765734758181776e-8f);
    return mrf_copysign(t, y);
}

/* ---------------------------------------------------------------------------------------------------------------- exp, exp2
 * n = rint(x log2 e), r = x - n ln2 (two parts, fma), e^r = 1 + r + r^2 P(r) on |r| <= ln2 / 2, result scaled by 2^n in two exact steps
 * (the second rounds once when the result is subnormal).  x > 88.7228394 -> +inf, x < -103.98 -> +0, NaN -> NaN. */
MRF_FN float mrf_scale2(float p, int n) {                        /* p * 2^n, n in [-160, 128], p in [1/2, 2] */
    const int n1 = n >> 1, n2 = n - n1;
    return (p * mrf_float((uint32_t)(n1 + 127) << 23)) * mrf_float((uint32_t)(n2 + 127) << 23);
}
MRF_FN float mrf_exp(float x) {
    if (x != x) return x;
    if (x > 88.7228394f) return mrf_inf();
    if (x < -103.98f) return 0.0f;
    const float fn = rintf(x * 1.44269502162933349609375f);
    float r = mrf_fma(fn, -0.693147182464599609375f, x);
    r = mrf_fma(fn, 1.90465421212593355e-9f, r);
    float p = mrf_fma(r, 1.9875691500e-4f, 1.3981999507e-3f);
    p = mrf_fma(p, r, 8.3334519073e-3f);
    p = mrf_fma(p, r, 4.1665795894e-2f);
    p = mrf_fma(p, r, 1.6666665459e-1f);
    p = mrf_fma(p, r, 5.0000001201e-1f);
    p = mrf_fma(r * r, p, r) + 1.0f;
    return mrf_scale2(p, (int)fn);
}
MRF_FN float mrf_exp2(float x) {
    if (x != x) return x;
    if (x >= 128.0f) return mrf_inf();
    if (x < -150.0f) return 0.0f;
    const float fn = rintf(x);
    const float r = x - fn;                                      /* exact */
    float p = mrf_fma(r, 1.535336188319500e-4f, 1.339887440266574e-3f);
    p = mrf_fma(p, r, 9.618437357674640e-3f);
    p = mrf_fma(p, r, 5.550332471162809e-2f);
    p = mrf_fma(p, r, 2.402264791363012e-1f);
    p = mrf_fma(p, r, 6.931472028550421e-1f);
    p = mrf_fma(p, r, 1.0f);
    return mrf_scale2(p, (int)fn);
}
/* logistic function of the material field's output layer (nerf/render_helper.py:93-117: torch.sigmoid) */
MRF_FN float mrf_sigmoid(float x) { return MRF_DIV(1.0f, 1.0f + mrf_exp(-x)); }

/* ---------------------------------------------------------------------------------------------------------------- integer powers of x >= 0
 * pow(x, 5) of evalFresnelSchlick: x^4 with a single rounding (the error of x * x carried by fma), times x (<= 2 ulp).  x^(2^k) (mFactor's pow(., 8), the bilateral filter's pow(., 128)): k squarings of a
 * two-float value (h + l), the rounding error of every product recovered by fma — < 1 ulp while the result stays normal. */
MRF_FN float mrf_pow5(float x) {
    const float x2 = x * x, e = mrf_fma(x, x, -x2), t = x2 * e;      /* x^2 = x2 + e exactly */
    return mrf_fma(x2, x2, t + t) * x;                              /* x^4 rounded once, then one product */
}
MRF_FN float mrf_pow2k(float x, int k) {
    float h = x, l = 0.0f;
    for (int i = 0; i < k; i++) {
        const float hh = h * h;
        const float e = mrf_fma(h, h, -hh);
        const float ll = mrf_fma(h + h, l, e);
        h = hh + ll;
        l = ll - (h - hh);
    }
    return h;
}

#endif /* MIRRES_FMATH_H */
