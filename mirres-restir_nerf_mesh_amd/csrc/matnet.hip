// matnet.hip — material field: multi-resolution hash grid (tcnn HashGrid semantics) + bias-free 32-32-32-6 ReLU MLP +
// sigmoid / affine range (MLPTexture3D.sample / sample_no_di, nerf/render_helper.py:93-117).
//
// tiny-cuda-nn is an un-vendored, un-pinned dependency of the reference (readme.md:30); its published algorithm is
// implemented here: 16 levels x 2 features, T = 2^19, base 16, per-level scale exp(ln(256)/15); levels 0-4 dense,
// 5-15 hashed with primes (1, 2654435761, 805459861); pos = fmaf(scale, x, 0.5); fp16 table, fp16 accumulation of the
// 8-corner interpolation; features level-major. The MLP runs in fp32 exactly like the reference's torch.nn.Linear
// stack: each output is a k-ordered fmaf chain, which is also the exact semantics of v_mfma_f32_32x32x2_f32
// (MI355X guide: "bit-for-bit a k-ordered f32 fmaf chain"), so the MFMA-tiled kernel and this per-lane kernel agree
// bitwise.
#include "engine.hpp"
#include "device_math.hpp"
#include <hip/hip_fp16.h>
#include <cmath>

namespace mr {

#define MR_BLOCK 256
#define MR_LEVELS 16

struct GridLevels { float scale[MR_LEVELS]; uint32_t res[MR_LEVELS]; uint32_t size[MR_LEVELS]; uint32_t offset[MR_LEVELS]; };

static GridLevels host_levels(uint32_t* total) {
    GridLevels L;
    const float per_level_scale = 1.4472692012786865f;  // fp32(exp(log(4096/16)/15)), render_helper.py:64-76
    const float log2_pls = log2f(per_level_scale);
    uint32_t offset = 0;
    for (int i = 0; i < MR_LEVELS; i++) {
        float scale = exp2f(i * log2_pls) * 16 - 1.0f;
        uint32_t res = (uint32_t)ceilf(scale) + 1;
        uint64_t dense = (uint64_t)res * res * res;
        uint32_t params = dense > 0x7fffffffull ? 0x7fffffffu : (uint32_t)dense;
        params = (params + 7u) / 8u * 8u;
        if (params > (1u << 19)) params = 1u << 19;
        L.scale[i] = scale; L.res[i] = res; L.size[i] = params; L.offset[i] = offset;
        offset += params;
    }
    if (total) *total = offset;
    return L;
}

MR_DEV uint32_t grid_index(uint32_t size, uint32_t res, uint32_t px, uint32_t py, uint32_t pz) {
    uint32_t stride = 1, index = 0;
    if (stride <= size) { index += px * stride; stride *= res; }
    if (stride <= size) { index += py * stride; stride *= res; }
    if (stride <= size) { index += pz * stride; stride *= res; }
    if (size < stride) index = (px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u);
    return index % size;
}

// encode one point (already normalised to [0,1]^3) into 32 fp16 features
MR_DEV void encode_point(const GridLevels& L, const __half2* __restrict__ grid, const float x[3], __half enc[32]) {
#pragma unroll 1
    for (int lv = 0; lv < MR_LEVELS; lv++) {
        const float scale = L.scale[lv]; const uint32_t res = L.res[lv], size = L.size[lv];
        const __half2* __restrict__ g = grid + L.offset[lv];
        float pos[3]; uint32_t pg[3];
#pragma unroll
        for (int d = 0; d < 3; d++) { float p = fmaf(scale, x[d], 0.5f); float fl = floorf(p); pg[d] = (uint32_t)(int)fl; pos[d] = p - fl; }
        __half r0 = __float2half(0.f), r1 = __float2half(0.f);
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            float w = 1.f; uint32_t pl[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; } else { w *= pos[d]; pl[d] = pg[d] + 1; }
            }
            const __half2 v = g[grid_index(size, res, pl[0], pl[1], pl[2])];
            r0 = __hadd(r0, __float2half(w * __low2float(v)));
            r1 = __hadd(r1, __float2half(w * __high2float(v)));
        }
        enc[2 * lv] = r0; enc[2 * lv + 1] = r1;
    }
}

struct MatNetD { const __half2* grid; const float *w0, *w1, *w2; float aabb_min[3], aabb_max[3], mn[6], mx[6]; };

// weights staged once per block in LDS (transposed to [k][o] so that the lanes of a wave read the same word: broadcast)
MR_DEV void stage_weights(const MatNetD& M, float* sw0, float* sw1, float* sw2) {
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) { sw0[i] = M.w0[i]; sw1[i] = M.w1[i]; }
    for (int i = threadIdx.x; i < 192; i += blockDim.x) sw2[i] = M.w2[i];
    __syncthreads();
}
MR_DEV void mlp_point(const float* sw0, const float* sw1, const float* sw2, const MatNetD& M, const __half enc[32], float out[6]) {
    float a[32], h[32];
#pragma unroll
    for (int i = 0; i < 32; i++) a[i] = __half2float(enc[i]);
#pragma unroll 4
    for (int o = 0; o < 32; o++) { float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k++) acc = fmaf(a[k], sw0[o * 32 + k], acc);
        h[o] = fmaxf(acc, 0.f); }
#pragma unroll 4
    for (int o = 0; o < 32; o++) { float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k++) acc = fmaf(h[k], sw1[o * 32 + k], acc);
        a[o] = fmaxf(acc, 0.f); }
#pragma unroll
    for (int o = 0; o < 6; o++) { float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k++) acc = fmaf(a[k], sw2[o * 32 + k], acc);
        float s = 1.0f / (1.0f + expf(-acc));
        out[o] = s * (M.mx[o] - M.mn[o]) + M.mn[o]; }
}
MR_DEV void normalise_pos(const MatNetD& M, const float* __restrict__ pos, size_t i, float x[3]) {
#pragma unroll
    for (int d = 0; d < 3; d++) x[d] = fminf(fmaxf((pos[3 * i + d] - M.aabb_min[d]) / (M.aabb_max[d] - M.aabb_min[d]), 0.f), 1.f);
}

__global__ void __launch_bounds__(MR_BLOCK) k_matnet_fwd(MatNetD M, GridLevels L, const float* __restrict__ pos, int n, float* __restrict__ out,
                                                         uint16_t* __restrict__ enc_out) {
    __shared__ float sw0[1024], sw1[1024], sw2[192];
    stage_weights(M, sw0, sw1, sw2);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[3]; normalise_pos(M, pos, i, x);
    __half enc[32];
    encode_point(L, M.grid, x, enc);
    if (enc_out) {
#pragma unroll
        for (int k = 0; k < 32; k++) enc_out[32 * (size_t)i + k] = __half_as_ushort(enc[k]);
    }
    float o[6]; mlp_point(sw0, sw1, sw2, M, enc, o);
#pragma unroll
    for (int k = 0; k < 6; k++) out[6 * (size_t)i + k] = o[k];
}

// renderer_restir.py:398-408 / 428-438 fused: hit pixels get kd = out[0:3], (roughness, metallic) = out[4:6]
__global__ void __launch_bounds__(MR_BLOCK) k_matnet_scatter(MatNetD M, GridLevels L, const float* __restrict__ occ, const float* __restrict__ pos, int n,
                                                             float* __restrict__ kd, float* __restrict__ rm, int use_scale, float sx, float sy, float sz,
                                                             int use_const, float c0, float c1, float c2, float c3, float c4) {
    __shared__ float sw0[1024], sw1[1024], sw2[192];
    if (!use_const) stage_weights(M, sw0, sw1, sw2);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (occ[i] >= 0.5f) {
        float o[6] = {c0, c1, c2, 0.f, c3, c4};
        if (!use_const) { float x[3]; normalise_pos(M, pos, i, x); __half enc[32]; encode_point(L, M.grid, x, enc); mlp_point(sw0, sw1, sw2, M, enc, o); }
        if (use_scale) { o[0] = o[0] * sx; o[1] = o[1] * sy; o[2] = o[2] * sz; }
        kd[3 * (size_t)i] = o[0]; kd[3 * (size_t)i + 1] = o[1]; kd[3 * (size_t)i + 2] = o[2];
        rm[2 * (size_t)i] = o[4]; rm[2 * (size_t)i + 1] = o[5];
    }
    if (use_scale) {  // torch.clamp(new_diffuse_map, 0, 1) over the whole map (:408)
#pragma unroll
        for (int k = 0; k < 3; k++) kd[3 * (size_t)i + k] = fminf(fmaxf(kd[3 * (size_t)i + k], 0.f), 1.f);
    }
}

__global__ void __launch_bounds__(MR_BLOCK) k_pack_grid(const float* __restrict__ in, uint16_t* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = __half_as_ushort(__float2half(in[i]));
}

static MatNetD matd(const mirres_matnet_t* m) {
    MatNetD M; M.grid = reinterpret_cast<const __half2*>(m->grid_f16); M.w0 = m->w0; M.w1 = m->w1; M.w2 = m->w2;
    for (int i = 0; i < 3; i++) { M.aabb_min[i] = m->aabb_min[i]; M.aabb_max[i] = m->aabb_max[i]; }
    for (int i = 0; i < 6; i++) { M.mn[i] = m->out_min[i]; M.mx[i] = m->out_max[i]; }
    return M;
}

int launch_matnet_scatter(const mirres_matnet_t* m, const float* occ, const float* pos, int n, float* kd, float* rm, int use_scale, const float* scale3,
                          const float* const_kd, const float* const_rm, hipStream_t s) {
    GridLevels L = host_levels(nullptr);
    MatNetD M; int use_const = 0; float c[5] = {0, 0, 0, 0, 0};
    if (m) M = matd(m);
    else { use_const = 1; M = MatNetD(); c[0] = const_kd[0]; c[1] = const_kd[1]; c[2] = const_kd[2]; c[3] = const_rm[0]; c[4] = const_rm[1]; }
    float sx = scale3 ? scale3[0] : 1.f, sy = scale3 ? scale3[1] : 1.f, sz = scale3 ? scale3[2] : 1.f;
    k_matnet_scatter<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, s>>>(M, L, occ, pos, n, kd, rm, use_scale, sx, sy, sz, use_const, c[0], c[1], c[2], c[3], c[4]);
    MR_LAUNCH_CHECK("matnet_scatter");
    return 0;
}

}  // namespace mr

using namespace mr;

extern "C" {

int mirres_matnet_grid_entries(void) { uint32_t t = 0; host_levels(&t); return (int)t; }

int mirres_matnet_pack_grid(const float* params_f32, uint16_t* grid_f16, int64_t n, void* stream) {
    if (!params_f32 || !grid_f16 || n < 0) { set_error("mirres_matnet_pack_grid: bad argument"); return MIRRES_E_ARG; }
    k_pack_grid<<<2048, MR_BLOCK, 0, (hipStream_t)stream>>>(params_f32, grid_f16, n);
    MR_LAUNCH_CHECK("matnet_pack_grid");
    return MIRRES_OK;
}

int mirres_matnet_fwd(const mirres_matnet_t* m, const float* pos, int n, float* out, uint16_t* enc_out, void* stream) {
    if (!m || !pos || !out || n < 0) { set_error("mirres_matnet_fwd: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    k_matnet_fwd<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(matd(m), host_levels(nullptr), pos, n, out, enc_out);
    MR_LAUNCH_CHECK("matnet_fwd");
    return MIRRES_OK;
}

int mirres_matnet_scatter(const mirres_matnet_t* m, const float* occ, const float* pos, int n, float* kd, float* rough_metal, int use_scale,
                          const float* h_scale3, void* stream) {
    if (!m || !occ || !pos || !kd || !rough_metal || n < 0) { set_error("mirres_matnet_scatter: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    return launch_matnet_scatter(m, occ, pos, n, kd, rough_metal, use_scale, h_scale3, nullptr, nullptr, (hipStream_t)stream);
}

}  // extern "C"
