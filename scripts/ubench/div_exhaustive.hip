// Exhaustive check of short correctly-rounded fp32 division sequences against the compiler's IEEE-754 division (v_div_scale / v_div_fmas /
// v_div_fixup sequence, 11 instructions) on gfx950: every pair of significands a, b in [1, 2) — 2^46 quotients, which covers all normal operands
// whose quotient is normal (scaling by powers of two commutes with every step) — for each candidate sequence.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math scripts/ubench/div_exhaustive.hip -o /tmp/div_ex && /tmp/div_ex [log2 of the b values to test, default 23 = all]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>

#define NV 4
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// A: reciprocal refined once (Newton), quotient corrected once                       rcp fma fma mul fma fma
__device__ __forceinline__ float div_a(float a, float b) {
    float r = rcp(b); float e = __builtin_fmaf(-b, r, 1.0f); r = __builtin_fmaf(e, r, r);
    float q = a * r; float m = __builtin_fmaf(-b, q, a); return __builtin_fmaf(m, r, q);
}
// B: raw reciprocal, quotient corrected twice                                         rcp mul fma fma fma fma
__device__ __forceinline__ float div_b(float a, float b) {
    float r = rcp(b); float q = a * r; float m = __builtin_fmaf(-b, q, a); q = __builtin_fmaf(m, r, q);
    m = __builtin_fmaf(-b, q, a); return __builtin_fmaf(m, r, q);
}
// C: the compiler's core without the scaling and fix-up instructions                rcp fma fma mul fma fma fma fma
__device__ __forceinline__ float div_c(float a, float b) {
    float r = rcp(b); float e = __builtin_fmaf(-b, r, 1.0f); r = __builtin_fmaf(e, r, r);
    float q = a * r; float m = __builtin_fmaf(-b, q, a); q = __builtin_fmaf(m, r, q);
    m = __builtin_fmaf(-b, q, a); return __builtin_fmaf(m, r, q);
}
// D: raw reciprocal, quotient corrected once                                          rcp mul fma fma
__device__ __forceinline__ float div_d(float a, float b) {
    float r = rcp(b); float q = a * r; float m = __builtin_fmaf(-b, q, a); return __builtin_fmaf(m, r, q);
}

__global__ void __launch_bounds__(256) k_check(uint32_t b0, uint32_t a_lo, uint32_t a_hi, unsigned long long* bad, uint32_t* first) {
    const uint32_t bm = b0 + blockIdx.x * blockDim.x + threadIdx.x;
    const float b = __uint_as_float(0x3f800000u | bm);
    unsigned long long n[NV] = {0, 0, 0, 0};
    for (uint32_t am = a_lo; am < a_hi; am++) {
        const float a = __uint_as_float(0x3f800000u | am);
        const float ref = a / b;
        const float v[NV] = {div_a(a, b), div_b(a, b), div_c(a, b), div_d(a, b)};
#pragma unroll
        for (int i = 0; i < NV; i++)
            if (__float_as_uint(v[i]) != __float_as_uint(ref)) {
                if (n[i] == 0) { const uint32_t s = atomicAdd(&first[i * 64], 1u); if (s < 20) { first[i * 64 + 1 + 2 * s] = am; first[i * 64 + 2 + 2 * s] = bm; } }
                n[i]++;
            }
    }
#pragma unroll
    for (int i = 0; i < NV; i++) if (n[i]) atomicAdd(&bad[i], n[i]);
}

// ---- the shipped forms (engine.hpp mr_div / mr_sqrt): variant A + v_div_fixup_f32 for zero / infinity / NaN operands and out-of-range quotients
__device__ __forceinline__ float div_ship(float a, float b) { return __builtin_amdgcn_div_fixupf(div_a(a, b), b, a); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// S1: y = rsq(x); s = x*y; one residual correction with h = y/2; zero / infinity passed through      rsq mul mul fma fma (+ class select)
__device__ __forceinline__ float sqrt_s1(float x) {
    float y = rsq(x); float s = x * y; float h = 0.5f * y; float r = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(r, h, s);
    return __builtin_amdgcn_classf(x, 0x2f0) ? x : s;      // +-0, +inf -> x (as the compiler's sequence does); +-denormal -> x (instead of a NaN)
}
// S2: S1 with a second residual correction
__device__ __forceinline__ float sqrt_s2(float x) {
    float y = rsq(x); float s = x * y; float h = 0.5f * y; float r = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(r, h, s);
    r = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(r, h, s);
    return __builtin_amdgcn_classf(x, 0x260) ? x : s;
}
// S3: hardware sqrt (1 ulp) + one residual correction with rsq/2
__device__ __forceinline__ float sqrt_s3(float x) {
    float s = __builtin_amdgcn_sqrtf(x); float h = 0.5f * rsq(x); float r = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(r, h, s);
    return __builtin_amdgcn_classf(x, 0x260) ? x : s;
}
__global__ void k_sqrt(unsigned long long* bad, uint32_t* first) {      // all 2^24 significand / exponent-parity combinations: x in [1, 4)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(0x3f800000u + i);
    const float ref = sqrtf(x);
    const float v[3] = {sqrt_s1(x), sqrt_s2(x), sqrt_s3(x)};
    for (int k = 0; k < 3; k++) if (__float_as_uint(v[k]) != __float_as_uint(ref)) { atomicAdd(&bad[k], 1ull); first[k] = i; }
}
// special values and the whole exponent range: class counts of mismatches of the shipped forms against the compiler's operations
__device__ __forceinline__ int expo(float x) { return (int)((__float_as_uint(x) >> 23) & 255u) - 127; }
__global__ void k_special(const float* vals, int n, unsigned long long* out, float* ex) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n) return;
    const float a = vals[i / n], b = vals[i % n];
    const float ref = a / b, got = div_ship(a, b);
    const bool same = __float_as_uint(ref) == __float_as_uint(got) || (ref != ref && got != got);
    if (!same) {
        const bool special = a == 0.f || b == 0.f || a != a || b != b || fabsf(a) == INFINITY || fabsf(b) == INFINITY;
        const int ea = expo(a), eb = expo(b), er = expo(ref);
        const bool safe = !special && abs(ea) <= 100 && abs(eb) <= 100 && er >= -100 && er <= 100;   // (er of 0 / inf / denormal results is outside)
        const int cls = special ? 0 : (safe ? 1 : 2);
        if (cls == 2) {
            const int ext = max(max(abs(ea), abs(eb)), abs(er));
            atomicMin(&out[4], (unsigned long long)ext);                                   // the least extreme exponent at which a mismatch occurs
            const bool fin = fabsf(ref) < INFINITY && fabsf(got) < INFINITY;
            const long long d = (long long)(__float_as_uint(ref) & 0x7fffffffu) - (long long)(__float_as_uint(got) & 0x7fffffffu);
            if (!fin || ref != ref || got != got || d > 1 || d < -1 || ((__float_as_uint(ref) ^ __float_as_uint(got)) >> 31)) atomicAdd(&out[5], 1ull);   // worse than one ulp
        }
        const unsigned long long s = atomicAdd(&out[cls], 1ull);
        if (s < 4) { ex[(cls * 4 + s) * 4 + 0] = a; ex[(cls * 4 + s) * 4 + 1] = b; ex[(cls * 4 + s) * 4 + 2] = ref; ex[(cls * 4 + s) * 4 + 3] = got; }
    }
    if (i < n) {          // sqrt of every value as well
        const float x = vals[i]; const float r0 = sqrtf(x), r1 = sqrt_s1(x);
        if (!(__float_as_uint(r0) == __float_as_uint(r1) || (r0 != r0 && r1 != r1))) { if (abs(expo(x)) <= 100) atomicAdd(&out[6], 1ull); const unsigned long long s = atomicAdd(&out[3], 1ull); if (s < 4) { ex[48 + s * 2] = x; ex[48 + s * 2 + 1] = r1; } }
    }
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    {   // square roots: exhaustive; special values and exponent sweep of the shipped forms
        unsigned long long* sb; uint32_t* sf; hipMalloc(&sb, 64); hipMalloc(&sf, 64); hipMemset(sb, 0, 64); hipMemset(sf, 0, 64);
        k_sqrt<<<(1u << 24) / 256, 256>>>(sb, sf); hipDeviceSynchronize();
        unsigned long long h[3]; hipMemcpy(h, sb, 24, hipMemcpyDeviceToHost);
        printf("sqrt over all 2^24 x in [1,4): mismatches vs sqrtf  S1 (rsq, 1 correction) %llu   S2 (rsq, 2 corrections) %llu   S3 (v_sqrt + correction) %llu\n", h[0], h[1], h[2]);
        static float vals[8192]; int n = 0;
        const uint32_t sp[] = {0x00000000u, 0x00000001u, 0x00400000u, 0x007fffffu, 0x00800000u, 0x00800001u, 0x3f800000u, 0x3fc00000u, 0x3fffffffu, 0x7f7fffffu, 0x7f800000u, 0x7fc00000u};
        for (uint32_t u : sp) { float f; memcpy(&f, &u, 4); vals[n++] = f; uint32_t m = u | 0x80000000u; memcpy(&f, &m, 4); vals[n++] = f; }
        uint32_t rng = 12345u;
        for (int e = 1; e < 255; e++) for (int k = 0; k < 12; k++) { rng = rng * 1664525u + 1013904223u; uint32_t u = ((uint32_t)e << 23) | (k == 0 ? 0u : k == 1 ? 0x7fffffu : (rng >> 9)) | ((k & 1) ? 0x80000000u : 0u); float f; memcpy(&f, &u, 4); vals[n++] = f; }
        float* dv; unsigned long long* out; float* ex; hipMalloc(&dv, n * 4); hipMalloc(&out, 64); hipMalloc(&ex, 64 * 4); hipMemset(out, 0, 64); hipMemset(ex, 0, 256); { unsigned long long big = 1000; hipMemcpy(out + 4, &big, 8, hipMemcpyHostToDevice); }
        hipMemcpy(dv, vals, n * 4, hipMemcpyHostToDevice);
        k_special<<<(n * n + 255) / 256, 256>>>(dv, n, out, ex); hipDeviceSynchronize();
        unsigned long long o[7]; float e[64]; hipMemcpy(o, out, 56, hipMemcpyDeviceToHost); hipMemcpy(e, ex, 256, hipMemcpyDeviceToHost);
        printf("division A + v_div_fixup over %d x %d operands (zeros, denormals, infinities, NaN, every exponent): mismatches vs IEEE — zero/inf/NaN operands %llu; normal operands and quotient with |exponent| <= 100: %llu; outside that range: %llu;  sqrt S1 on the same values: %llu, of which with |exponent| <= 100: %llu\n", n, n, o[0], o[1], o[2], o[3], o[6]);
        printf("   outside: least extreme exponent with a mismatch %llu; mismatches worse than one ulp (or non-finite) %llu\n", o[4], o[5]);
        for (int c = 0; c < 3; c++) for (int k = 0; k < 4 && k < (int)o[c]; k++) printf("   class %d: %g / %g = %g, got %g\n", c, e[(c * 4 + k) * 4], e[(c * 4 + k) * 4 + 1], e[(c * 4 + k) * 4 + 2], e[(c * 4 + k) * 4 + 3]);
        for (int k = 0; k < 4 && k < (int)o[3]; k++) printf("   sqrt(%g) got %g\n", e[48 + 2 * k], e[48 + 2 * k + 1]);
    }
    const int lb = argc > 1 ? atoi(argv[1]) : 23;          // 2^lb significands of b (evenly spread when lb < 23, plus the all-ones significand)
    unsigned long long* bad; uint32_t* first;
    hipMalloc(&bad, NV * 8); hipMalloc(&first, NV * 64 * 4); hipMemset(bad, 0, NV * 8); hipMemset(first, 0, NV * 64 * 4);
    const uint32_t per = 1u << 18;                          // b values per launch
    const uint32_t launches = (1u << lb) / per ? (1u << lb) / per : 1;
    const uint32_t stride = (1u << 23) / launches;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    for (uint32_t l = 0; l < launches; l++) {
        const uint32_t b0 = (lb == 23) ? l * per : (l * stride + (l * 2654435761u) % (stride - per + 1));
        k_check<<<per / 256, 256>>>(b0, 0u, 1u << 23, bad, first);
        if ((l & 15) == 15 || l + 1 == launches) { hipDeviceSynchronize(); unsigned long long h[NV]; hipMemcpy(h, bad, NV * 8, hipMemcpyDeviceToHost); printf("launch %u/%u mismatches A %llu B %llu C %llu D %llu\n", l + 1, launches, h[0], h[1], h[2], h[3]); }
    }
    k_check<<<1, 256>>>((1u << 23) - 256, 0u, 1u << 23, bad, first);       // the last 256 significands incl. all ones (always)
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[NV]; uint32_t f[NV * 64]; hipMemcpy(h, bad, NV * 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, NV * 64 * 4, hipMemcpyDeviceToHost);
    const char* names[NV] = {"A rcp+newton, 1 correction (6 instr)", "B raw rcp, 2 corrections (6 instr)", "C compiler core, no scale/fixup (8 instr)", "D raw rcp, 1 correction (4 instr)"};
    printf("pairs tested: %.4g in %.1f s\n", (double)(launches * (double)per + 256) * (double)(1u << 23), ms / 1e3);
    for (int i = 0; i < NV; i++) {
        printf("%-44s mismatches vs IEEE division: %llu", names[i], h[i]);
        const uint32_t k = f[i * 64] < 4 ? f[i * 64] : 4;
        for (uint32_t s = 0; s < k; s++) printf("  (a=0x%06x b=0x%06x)", f[i * 64 + 1 + 2 * s], f[i * 64 + 2 + 2 * s]);
        printf("\n");
    }
    return 0;
}
