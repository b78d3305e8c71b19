// Dev micro-benchmark (GPU box): what a wave pays for 64 scattered 64-byte records — one record per LANE (four dwordx4 gathers, every lane on a line of its own:
// what k_spatial_resolve / k_spatial_gen / k_bounce_gen do for neighbour pixels) against one record per QUAD of lanes (lane l loads quarter l % 4 of the record of
// "pixel" 16 k + l / 4: four instructions, each touching 16 lines instead of 64), with and without the LDS transpose that gives every lane its own record back.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/gather_coop.hip -o /tmp/gather_coop && /tmp/gather_coop
// The index pattern imitates the spatial pass: record of pixel p = a pixel within +-30 of p in a 1600-wide image (a different one per lane and pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ int neighbour(int p, int pass, int W, int H) {
    const uint32_t h = hash32((uint32_t)p * 9781u + (uint32_t)pass * 6271u);
    int x = p % W + (int)(h % 61u) - 30, y = p / W + (int)((h >> 8) % 61u) - 30;
    x = x < 0 ? 0 : (x >= W ? W - 1 : x); y = y < 0 ? 0 : (y >= H ? H - 1 : y);
    return y * W + x;
}
// tile mapping as the resolve kernel: one wave per 8 x 8 pixel tile
__device__ __forceinline__ int tile_pixel(int W) { const int tiles_x = W / 8; const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x; return (ty * 8 + (threadIdx.x >> 3)) * W + tx * 8 + (threadIdx.x & 7); }

template <int MODE>   // 0: per lane; 1: per quad, no transpose (each lane keeps what it loaded: lower bound); 2: per quad + LDS transpose
__global__ void __launch_bounds__(64) k_gather(const float4* __restrict__ rec, int W, int H, int passes, float* __restrict__ out) {
    __shared__ float4 stage[64 * 4];
    const int p = tile_pixel(W);
    const int lane = threadIdx.x;
    float acc = 0.f;
    for (int pass = 0; pass < passes; pass++) {
        const int q = neighbour(p, pass, W, H);
        if (MODE == 0) {
            const float4 a = rec[4 * (size_t)q], b = rec[4 * (size_t)q + 1], c = rec[4 * (size_t)q + 2], d = rec[4 * (size_t)q + 3];
            acc += a.x + b.y + c.z + d.w;
        } else {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int src = 16 * k + (lane >> 2);                  // whose record this lane helps to load
                const int qs = __shfl(q, src, 64);
                v[k] = rec[4 * (size_t)qs + (lane & 3)];
            }
            if (MODE == 1) acc += v[0].x + v[1].y + v[2].z + v[3].w;
            else {
#pragma unroll
                for (int k = 0; k < 4; k++) stage[(16 * k + (lane >> 2)) * 4 + (lane & 3)] = v[k];
                __syncthreads();
                const float4 a = stage[lane * 4], b = stage[lane * 4 + 1], c = stage[lane * 4 + 2], d = stage[lane * 4 + 3];
                __syncthreads();
                acc += a.x + b.y + c.z + d.w;
            }
        }
    }
    out[p] = acc;
}

int main() {
    const int W = 1600, H = 1600, N = W * H, passes = 5;
    float4* rec; float* out;
    CHECK(hipMalloc(&rec, sizeof(float4) * 4 * (size_t)N)); CHECK(hipMalloc(&out, sizeof(float) * N));
    std::vector<float> h(16 * (size_t)N);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761ull) & 1023) * 0.001f;
    CHECK(hipMemcpy(rec, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = (W / 8) * (H / 8);
    std::vector<float> ref(N), got(N);
    for (int mode = 0; mode < 3; mode++) {
        auto launch = [&]() {
            if (mode == 0) k_gather<0><<<grid, 64>>>(rec, W, H, passes, out);
            else if (mode == 1) k_gather<1><<<grid, 64>>>(rec, W, H, passes, out);
            else k_gather<2><<<grid, 64>>>(rec, W, H, passes, out);
        };
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); for (int r = 0; r < 10; r++) launch(); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        CHECK(hipMemcpy(got.data(), out, sizeof(float) * N, hipMemcpyDeviceToHost));
        if (mode == 0) ref = got;
        size_t bad = 0; if (mode == 2) for (int i = 0; i < N; i++) bad += got[i] != ref[i];
        printf("mode %d (%s): %.1f us per launch, %d x %d records of 64 B = %.2f TB/s of records%s\n", mode,
               mode == 0 ? "one record per lane, 4 dwordx4 gathers" : (mode == 1 ? "one record per quad of lanes, no transpose (lower bound)" : "one record per quad of lanes + LDS transpose"),
               ms * 1e3, passes, N, passes * (double)N * 64 / (ms * 1e-3) / 1e12, mode == 2 ? (bad ? "  MISMATCH" : "  (same sums as mode 0)") : "");
    }
    return 0;
}
