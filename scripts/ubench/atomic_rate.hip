// Dev micro-benchmark (GPU box): how many fp32 atomic adds per second MI355X takes, by memory scope of the atomic.
// The gradient scatters of the training step (hash-grid table: k_matnet_bwd; environment map: k_direct_bwd) run at ~16 G atomics/s with TCC_EA0_ATOMIC == TCC_ATOMIC
// (profiles/r06_pmc_train.txt): every device-scope atomic leaves the XCD's L2 for the memory side — it has to, the eight L2s are not coherent with each other.
// An atomic of WORKGROUP scope may be executed in the XCD's own L2. That is only correct when no other XCD touches the word: one private copy of the table per XCD
// (index = HW_REG_XCC_ID), summed afterwards.  This program measures both, on scattered and on contended addresses, and checks the sums.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// MODE 0: agent scope, one table. 1: workgroup scope, table copy of this XCD. 2: agent scope, table copy of this XCD (same addresses as 1, device-scope instruction)
template <int MODE>
__global__ void __launch_bounds__(256) k_add(float* table, uint32_t words, uint32_t per_thread, uint32_t span) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float* base = table + (MODE == 0 ? 0 : (size_t)xcc_id() * words);
    for (uint32_t i = 0; i < per_thread; i++) {
        const uint32_t a = hash32(t * per_thread + i) % span;
        if (MODE == 1) __hip_atomic_fetch_add(base + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(base + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// GROUP consecutive lanes add to GROUP consecutive dwords of one random (GROUP * 4)-byte aligned segment: do the lanes of one instruction that fall into the same
// 32-byte sector travel as ONE request? (the channels of an environment texel, the two features of a hash-grid entry)
template <int GROUP>
__global__ void __launch_bounds__(256) k_add_grouped(float* table, uint32_t words, uint32_t per_thread) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < per_thread; i++) {
        const uint32_t seg = hash32((t / GROUP) * per_thread + i) % (words / GROUP);
        __hip_atomic_fetch_add(table + (size_t)seg * GROUP + (t % GROUP), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_sum(const float* table, size_t n, double* out) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += table[i];
    atomicAdd(out, acc);
}
int main() {
    const uint32_t words = 12600000;      // the hash grid's gradient table: 12.6 M floats
    float* table; CHECK(hipMalloc(&table, sizeof(float) * (size_t)words * 8));
    double* d_sum; CHECK(hipMalloc(&d_sum, 8));
    const int blocks = 256 * 16, per_thread = 64;
    const double total = (double)blocks * 256 * per_thread;
    const char* names[3] = {"agent scope, one table", "workgroup scope, per-XCD copy", "agent scope, per-XCD copy"};
    for (uint32_t span : {words, 131072u * 3u, 4096u, 64u}) {
        printf("---- %u distinct words (%.1f M atomics per launch)\n", span, total / 1e6);
        for (int mode = 0; mode < 3; mode++) {
            CHECK(hipMemset(table, 0, sizeof(float) * (size_t)words * 8)); CHECK(hipMemset(d_sum, 0, 8));
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto launch = [&] { if (mode == 0) k_add<0><<<blocks, 256>>>(table, words, per_thread, span); else if (mode == 1) k_add<1><<<blocks, 256>>>(table, words, per_thread, span); else k_add<2><<<blocks, 256>>>(table, words, per_thread, span); };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(table, 0, sizeof(float) * (size_t)words * 8));
            hipEventRecord(a); launch(); hipEventRecord(b); CHECK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            k_sum<<<1024, 256>>>(table, (size_t)words * 8, d_sum); double s; CHECK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
            printf("  %-32s %8.3f ms  %7.2f G atomics/s   sum %.0f (%s)\n", names[mode], ms, total / ms / 1e6, s, s == total ? "exact" : "WRONG");
        }
    }
    printf("---- grouped lanes: G consecutive lanes on G consecutive dwords of one random aligned segment of the %u-word table\n", words);
    for (int G : {1, 2, 4, 8}) {
        CHECK(hipMemset(table, 0, sizeof(float) * (size_t)words)); CHECK(hipMemset(d_sum, 0, 8));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        auto launch = [&] { if (G == 1) k_add_grouped<1><<<blocks, 256>>>(table, words, per_thread); else if (G == 2) k_add_grouped<2><<<blocks, 256>>>(table, words, per_thread);
                            else if (G == 4) k_add_grouped<4><<<blocks, 256>>>(table, words, per_thread); else k_add_grouped<8><<<blocks, 256>>>(table, words, per_thread); };
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(table, 0, sizeof(float) * (size_t)words));
        hipEventRecord(a); launch(); hipEventRecord(b); CHECK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        k_sum<<<1024, 256>>>(table, (size_t)words, d_sum); double s; CHECK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
        printf("  G = %d  %8.3f ms  %7.2f G lane-atomics/s   sum %.0f (%s)\n", G, ms, total / ms / 1e6, s, s == total ? "exact" : "WRONG");
    }
    return 0;
}
