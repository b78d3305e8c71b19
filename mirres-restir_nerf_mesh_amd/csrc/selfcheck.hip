// selfcheck.hip — mirres_selfcheck_arith: the short division / square-root sequences of the shading kernels (device_math.hpp under MR_LEAN_FP)
// against the compiler's IEEE-754 operations, exhaustively over significands (include/mirres.h). Same header, same flags as passes.hip / shading.hip:
// what is checked is the code that ships, not a copy of it (scripts/ubench/div_exhaustive.hip is the stand-alone study that chose the sequences).
#ifndef MR_LEAN_FP
#define MR_LEAN_FP 1
#endif
#include "engine.hpp"
#include "device_math.hpp"

namespace mr {

__global__ void __launch_bounds__(256) k_check_div(uint32_t b0, unsigned long long* __restrict__ bad) {
    const uint32_t bm = b0 + blockIdx.x * blockDim.x + threadIdx.x;
    const float b = __uint_as_float(0x3f800000u | bm);
    unsigned long long n = 0;
    for (uint32_t am = 0; am < (1u << 23); am++) {
        const float a = __uint_as_float(0x3f800000u | am);
        n += __float_as_uint(mr_div(a, b)) != __float_as_uint(a / b) ? 1u : 0u;
    }
    if (n) atomicAdd(&bad[1], n);
}
__global__ void __launch_bounds__(256) k_check_rcp_sqrt(unsigned long long* __restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;       // 2^24 threads
    const float x = __uint_as_float(0x3f800000u + i);              // [1, 4): both exponent parities of the square root
    if (__float_as_uint(mr_sqrt(x)) != __float_as_uint(sqrtf(x))) atomicAdd(&bad[3], 1ull);
    if (i < (1u << 23) && __float_as_uint(mr_rcp(x)) != __float_as_uint(1.0f / x)) atomicAdd(&bad[2], 1ull);
}


// ---- include/mirres_fmath.h on the device (function ids as in include/mirres.h)
MR_DEV uint32_t pair_hash(uint32_t i) { i ^= i >> 16; i *= 0x7feb352du; i ^= i >> 15; i *= 0x846ca68bu; i ^= i >> 16; return i; }
MR_DEV float fmath_eval(int fn, float a, float b) {
    switch (fn) {
        case 0: return mrf_sin(a);
        case 1: return mrf_cos(a);
        case 2: return mrf_acos(a);
        case 3: return mrf_exp(a);
        case 4: return mrf_exp2(a);
        case 5: return mrf_pow5(a);
        case 6: return mrf_pow2k(a, 3);
        case 7: return mrf_pow2k(a, 7);
        case 8: return mrf_sigmoid(a);
        case 16: return mrf_atan2(a, b);
        case 17: return mr_div(b, a);
        case 18: return mr_sqrt(a);
        default: return 0.f;
    }
}
__global__ void __launch_bounds__(256) k_fmath_eval(int fn, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fmath_eval(fn, a[i], b ? b[i] : 0.f);
}
// 2^20 blocks x 256 threads x 16 arguments per thread cover 2^32 arguments; one atomic per wave
__global__ void __launch_bounds__(256) k_fmath_checksum(int fn, uint32_t first, unsigned long long count, unsigned long long* __restrict__ out) {
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long sum = 0;
    for (unsigned long long i = t; i < count; i += nthreads) {
        const uint32_t u = first + (uint32_t)i;
        const uint32_t h = pair_hash(u);
        // fn 17 (short division): numerator magnitudes 2^-20 .. 2^20 (the sequence is proved for operands and quotients within 2^+-102)
        const float r = fmath_eval(fn, __uint_as_float(u), __uint_as_float(fn == 17 ? ((h & 0x807fffffu) | ((107u + (h >> 23) % 41u) << 23)) : h));
        const uint32_t rb = r != r ? 0x7fc00000u : __float_as_uint(r);
        sum += (unsigned long long)rb * (2ull * u + 1ull);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane_id() == 0) atomicAdd(out, sum);
}

}  // namespace mr

using namespace mr;

extern "C" int mirres_fmath_eval(int fn, const float* a, const float* b, float* out, long long n, void* stream) {
    if (!a || !out || n < 0 || !((fn >= 0 && fn <= 8) || (fn >= 16 && fn <= 18)) || (fn >= 16 && fn <= 17 && !b)) { set_error("mirres_fmath_eval: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    k_fmath_eval<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(fn, a, b, out, n);
    MR_LAUNCH_CHECK("fmath_eval");
    return MIRRES_OK;
}
extern "C" int mirres_fmath_checksum(int fn, unsigned int first, unsigned long long count, unsigned long long* out, void* stream) {
    if (!out || count > (1ull << 32) || !((fn >= 0 && fn <= 8) || (fn >= 16 && fn <= 18))) { set_error("mirres_fmath_checksum: bad argument"); return MIRRES_E_ARG; }
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* d = nullptr;
    MR_HIP(hipMalloc(&d, sizeof(unsigned long long)));
    MR_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), s));
    if (count) {
        const unsigned long long want = (count + 256ull * 16ull - 1) / (256ull * 16ull);
        k_fmath_checksum<<<(unsigned)(want < 1 ? 1 : (want > (1u << 20) ? (1u << 20) : want)), 256, 0, s>>>(fn, first, count, d);
        MR_LAUNCH_CHECK("fmath_checksum");
    }
    MR_HIP(hipMemcpyAsync(out, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    MR_HIP(hipStreamSynchronize(s));
    MR_HIP(hipFree(d));
    return MIRRES_OK;
}

extern "C" int mirres_selfcheck_arith(int log2_b, unsigned long long out[4], void* stream) {
    if (!out || log2_b < 8 || log2_b > 23) { set_error("mirres_selfcheck_arith: bad argument"); return MIRRES_E_ARG; }
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* d = nullptr;
    MR_HIP(hipMalloc(&d, 4 * sizeof(unsigned long long)));
    MR_HIP(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), s));
    const uint32_t per = log2_b < 16 ? (1u << log2_b) : (1u << 16);          // significands of b per launch
    const uint32_t launches = (1u << log2_b) / per, stride = (1u << 23) / launches;
    for (uint32_t l = 0; l < launches; l++) {
        const uint32_t b0 = log2_b == 23 ? l * per : l * stride + (uint32_t)(((unsigned long long)l * 2654435761ull) % (stride - per + 1));
        k_check_div<<<per / 256, 256, 0, s>>>(b0, d);
    }
    k_check_div<<<1, 256, 0, s>>>((1u << 23) - 256, d);                      // the last 256 significands (all ones among them), always
    k_check_rcp_sqrt<<<(1u << 24) / 256, 256, 0, s>>>(d);
    MR_LAUNCH_CHECK("selfcheck_arith");
    MR_HIP(hipMemcpyAsync(out, d, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    MR_HIP(hipStreamSynchronize(s));
    MR_HIP(hipFree(d));
    out[0] = ((unsigned long long)launches * per + 256ull) << 23;
    return MIRRES_OK;
}
