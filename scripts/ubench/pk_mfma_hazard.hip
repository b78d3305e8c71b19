// pk_mfma_hazard.hip — stand-alone reproducer attempt for the "packed fp32 next to MFMA" corruption of round 1 (csrc/build.py).
// Observation then: with the MFMA material-net kernel resident on a CU, waves of OTHER kernels that executed v_pk_mul/add/fma_f32 occasionally produced a wrong
// 16-lane group; the library has been built without packed fp32 since (-fno-slp-vectorize -fno-vectorize).  This program isolates the two ingredients:
//   * 512-thread blocks = 8 waves = 2 per SIMD: waves 0-3 issue dependent-free v_mfma back to back (f16 32x32x16 or f32 32x32x2, selectable),
//     waves 4-7 run v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 chains (inline asm: the instruction is what is tested, not the vectoriser) on known data and compare
//     EVERY lane and BOTH halves against the same chain in scalar v_fma_f32 / v_mul_f32 / v_add_f32, bit for bit (a packed fp32 op is specified as two IEEE ops);
//   * variants: MFMA waves present / absent, MFMA in a SECOND kernel on another stream (the production situation) / in the same kernel.
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_mfma scripts/ubench/pk_mfma_hazard.hip && /tmp/pk_mfma
// Output: mismatching (lane, half) results per variant.  0 everywhere = the hardware executes packed fp32 correctly beside MFMA in this isolated setting, and the
// round-1 corruption needs something else that the production kernels had (see DESIGN.md section 4 for what was then found).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ f32x16_t mfma_burst(int kind, int iters, float seed) {
    f32x16_t acc[4];
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) acc[q][r] = seed;
    half8_t a, b; for (int t = 0; t < 8; t++) { a[t] = (_Float16)(0.001f * (t + 1)); b[t] = (_Float16)(0.002f * (t + 1)); }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {      // four independent accumulators: the matrix pipe never waits
            if (kind == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(0.001f * (q + 1), 0.002f, acc[q], 0, 0, 0);
        }
    }
    return acc[0] + acc[1] + acc[2] + acc[3];
}
// packed chain against the scalar chain; returns the number of (half) results that differ in bits
__device__ __forceinline__ unsigned pk_check(int iters, uint32_t salt) {
    const int lane = threadIdx.x & 63;
    f32x2_t x, y, z; float sx0, sx1, sy0, sy1, sz0, sz1;
    x[0] = sx0 = 1.0f + 0.001f * lane + 1e-6f * (salt & 1023); x[1] = sx1 = 0.5f + 0.003f * lane;
    y[0] = sy0 = 0.999f - 1e-4f * lane;                        y[1] = sy1 = 1.0001f + 2e-5f * lane;
    z[0] = sz0 = 1e-3f * (lane + 1);                           z[1] = sz1 = -2e-3f * (lane + 1);
    unsigned bad = 0;
    for (int i = 0; i < iters; i++) {
        f32x2_t p, m, s;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(x), "v"(y), "v"(z));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(p), "v"(y));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(m), "v"(z));
        const float p0 = __builtin_fmaf(sx0, sy0, sz0), p1 = __builtin_fmaf(sx1, sy1, sz1);
        float m0, m1, s0, s1;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(p0), "v"(sy0)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(p1), "v"(sy1));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(m0), "v"(sz0)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(m1), "v"(sz1));
        bad += (__float_as_uint(s[0]) != __float_as_uint(s0)) + (__float_as_uint(s[1]) != __float_as_uint(s1));
        bad += (__float_as_uint(p[0]) != __float_as_uint(p0)) + (__float_as_uint(p[1]) != __float_as_uint(p1));
        x = s; sx0 = s0; sx1 = s1;                       // keeps values near 1: the chain does not overflow (y ~ 1, z small)
        x[0] = sx0 = sx0 * 0.5f + 0.5f; x[1] = sx1 = sx1 * 0.5f + 0.25f;
    }
    return bad;
}
// mode 0: pk waves only; 1: waves 0-3 MFMA f16 + waves 4-7 pk; 2: waves 0-3 MFMA f32 + waves 4-7 pk; 3: MFMA only (companion kernel for the two-stream variant)
__global__ void __launch_bounds__(512) k(int mode, int kind, int iters, unsigned long long* bad, float* sink) {
    const int wave = threadIdx.x >> 6;
    if (mode == 3 || ((mode == 1 || mode == 2) && wave < 4)) {
        f32x16_t r = mfma_burst(mode == 3 ? kind : mode - 1, iters * 4, 0.f);
        if (r[0] == 123.456f) sink[threadIdx.x] = r[3];      // keeps the MFMAs alive
    } else {
        const unsigned b = pk_check(iters, blockIdx.x * 8 + wave);
        if (b) atomicAdd(bad, (unsigned long long)b);
    }
}
int main() {
    unsigned long long* bad; float* sink;
    CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&sink, 4096));
    hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
    const int blocks = 256 * 4, iters = 20000;
    const char* names[] = {"packed fp32 alone", "packed fp32 beside v_mfma_f32_32x32x16_f16 (same kernel, 2 waves per SIMD)", "packed fp32 beside v_mfma_f32_32x32x2_f32 (same kernel)",
                           "packed fp32 kernel beside an f16-MFMA kernel on a second stream", "packed fp32 kernel beside an f32-MFMA kernel on a second stream"};
    for (int v = 0; v < 5; v++) {
        unsigned long long total = 0;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipMemset(bad, 0, 8));
            if (v < 3) k<<<blocks, 512, 0, s1>>>(v, 0, iters, bad, sink);
            else { k<<<blocks, 512, 0, s2>>>(3, v - 3, iters, bad, sink); k<<<blocks, 512, 0, s1>>>(0, 0, iters, bad, sink); k<<<blocks, 512, 0, s2>>>(3, v - 3, iters, bad, sink); }
            CHECK(hipDeviceSynchronize());
            unsigned long long h = 0; CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost)); total += h;
        }
        printf("%-90s mismatching results: %llu of %.3g\n", names[v], total, 5.0 * blocks * (v == 1 || v == 2 ? 256.0 : 512.0) * iters * 4);
    }
    return 0;
}
