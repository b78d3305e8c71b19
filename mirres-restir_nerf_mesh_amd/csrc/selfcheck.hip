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

}  // namespace mr

using namespace mr;

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
