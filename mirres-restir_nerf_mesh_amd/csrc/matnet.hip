// matnet.hip — material field: multi-resolution hash grid (tcnn HashGrid semantics) + bias-free 32-32-32-6 ReLU MLP +
// sigmoid / affine range (MLPTexture3D.sample / sample_no_di, nerf/render_helper.py:93-117).
//
// tiny-cuda-nn is an un-vendored, un-pinned dependency of the reference (readme.md:30); its published algorithm is
// implemented here: 16 levels x 2 features, T = 2^19, base 16, per-level scale exp(ln(256)/15); levels 0-4 dense,
// 5-15 hashed with primes (1, 2654435761, 805459861); pos = fmaf(scale, x, 0.5); fp16 table, fp16 accumulation of the
// 8-corner interpolation; features level-major. The MLP runs in fp32 like the reference's torch.nn.Linear stack: the
// per-lane kernels (k_matnet_fwd / k_matnet_scatter) evaluate each output as a k-ordered fmaf chain; the MFMA-tiled kernel
// (k_mlp_mfma, the production path) runs the SAME chains on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, K = 2 per step in ascending k): same bits.
#include "engine.hpp"
#include "device_math.hpp"
#include "device_grid.hpp"
#include <hip/hip_fp16.h>
#include <cmath>

namespace mr {

#define MR_BLOCK 256

// encode one point (already normalised to [0,1]^3) into 32 fp16 features
MR_DEV void encode_point(const GridLevels& L, const __half2* __restrict__ grid, const float x[3], __half enc[32]) {
#pragma unroll 1
    for (int lv = 0; lv < MR_LEVELS; lv++) {
        const float scale = L.scale[lv]; const uint32_t res = L.res[lv], size = L.size[lv];
        const __half2* __restrict__ g = grid + L.offset[lv];
        float pos[3]; uint32_t pg[3];
#pragma unroll
        for (int d = 0; d < 3; d++) { float p = fmaf(scale, x[d], 0.5f); float fl = floorf(p); pg[d] = (uint32_t)(int)fl; pos[d] = p - fl; }
        uint32_t ci[8];
        corner_indices(size, res, pg, ci);
        __half2 v[8];
        gather_cell(g, ci, (uint64_t)res * res * res > (uint64_t)size && (size & (size - 1u)) == 0u, v);
        __half2 r = __floats2half2_rn(0.f, 0.f);
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            float w = 1.f;
#pragma unroll
            for (int d = 0; d < 3; d++) w *= (idx & (1u << d)) == 0 ? 1 - pos[d] : pos[d];
            r = __hadd2(r, weighted_half2(w, v[idx]));
        }
        enc[2 * lv] = __low2half(r); enc[2 * lv + 1] = __high2half(r);
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
        float s = mrf_sigmoid(acc);
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


// ---------------------------------------------------------------- MFMA-tiled MLP (the one dense contraction of the path)
// LDS staging (MODE 1): feature rows padded to 40 halfs so that the 16-byte fragment reads of 32 consecutive points fall on distinct bank
// groups (MI355X guide, LDS section: b128 reads are served in 16-lane groups over 64 banks).
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define MR_FROW 40

// 64 points per wave on the fp32 matrix pipe: bit-equal to an fp32 fmaf chain.
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate): D[i][j] = fma(A[i][1], B[1][j], fma(A[i][0], B[0][j], C[i][j])) — an fmaf chain in k order, bit for bit
// (MI355X_MICROARCH.md, MFMA table; checked against the per-lane kernel in tests/test_gpu_matnet.py).  The reference evaluates the layers in fp32
// (torch.nn.Linear on fp32 activations, nerf/render_helper.py:28-51); with sixteen K = 2 steps per layer in ascending k the accumulator of output neuron i
// goes through exactly the chain  acc = fmaf(x[k], W[i][k], acc), k = 0 .. 31  of the per-lane kernel above — the material values, and with them every
// lobe choice at an indirect vertex, are the same bits whichever kernel (or host restatement) produced them.
// Layout: the product is computed transposed, C[neuron][point] = W X^T, so that a layer's accumulator turns into the next layer's B operand without
// leaving registers.  A = weights, lane (i = lane & 31, kh = lane >> 5) holds W[i][2 s + kh] of step s; B = activations, lane
// (kh, j) holds x[2 s + kh] of point j; D: lane (j, half) holds neurons (r & 3) + 8 (r >> 2) + 4 half in register r.  Between layers the B operand of step
// s needs neurons 2 s and 2 s + 1 of point j in the two halves of ONE register: v_permlane32_swap_b32 on the register pairs (r, r + 1), r even, turns
// D into exactly that — register r then holds neurons (n, n + 1), n = (r & 3) + 8 (r >> 2) + 4 (r & 1) - (r & 1), i.e. step s lives in register
// (s & ~3) | ((s & 1) << 1) | ((s >> 1) & 1).  Activations never leave registers; 8 swaps and 16 ReLUs per layer and tile.
// MODE 0: features given (enc_in fp16 [n,32]) -> out6[n,6];  MODE 1: positions of the compacted pixel list -> scatter kd / (rough, metal).
// NT = 32-point tiles per wave, processed in lock-step (NT independent accumulator chains).
// Cost: 48 MFMAs of 64 cycles per 32 points.  Round 2 ran this MLP on the f16 pipe with hi / lo split operands (32 MFMAs of 32 cycles per 64 points,
// 3e-6 from the fmaf chain): the GEMM phase alone took 57 us per 2.56 M points against 201 us now, the FRAME is unchanged (1075 vs 1077 Msamples/s:
// inside it the kernel is bound by its 128 hash-grid gathers per point) — and every frame with the material field is now bit-equal to the CPU
// restatement, which is what the exchange bought (profiles/r03_mlp_ab.txt).
MR_DEV int step_reg(int s) { return (s & ~3) | ((s & 1) << 1) | ((s >> 1) & 1); }
MR_DEV void swap_halves(float& a, float& b) {     // a.upper <-> b.lower
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
template <int MODE, int NT>
__global__ void __launch_bounds__(MR_BLOCK) k_mlp_mfma(MatNetD M, GridLevels L, const uint16_t* __restrict__ enc_in, const float* __restrict__ pos,
                                                         const int32_t* __restrict__ index, const uint32_t* __restrict__ d_count, int n_fixed,
                                                         float* __restrict__ out6, float* __restrict__ kd, float* __restrict__ rm, int use_scale,
                                                         float sx, float sy, float sz) {
    constexpr int PTS = (MR_BLOCK / 64) * NT * 32;   // points per block iteration
    __shared__ __attribute__((aligned(16))) _Float16 sF[MODE == 1 ? MR_BLOCK * MR_FROW : 8];
    __shared__ float sW[2240];
    const int n = d_count ? (int)*d_count : n_fixed;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    for (int i = threadIdx.x; i < 2240; i += MR_BLOCK) sW[i] = i < 1024 ? M.w0[i] : (i < 2048 ? M.w1[i - 1024] : M.w2[i - 2048]);   // coalesced
    __syncthreads();
    float wa[3][16];                                  // A fragments: W_l[j][2 s + half]
#pragma unroll
    for (int l = 0; l < 3; l++) {
#pragma unroll
        for (int st = 0; st < 16; st++) wa[l][st] = (l == 2 && j >= 6) ? 0.f : sW[l * 1024 + j * 32 + 2 * st + half];
    }
    int prow[NT];
#pragma unroll
    for (int mt = 0; mt < NT; mt++) prow[mt] = wave * NT * 32 + mt * 32 + j;
    half8_t f[NT][4];                                 // the 32 features of this lane's point (both halves of the wave hold the row; each picks its k parity)
    auto fetch = [&](int base_) {
#pragma unroll
        for (int mt = 0; mt < NT; mt++) {
            const int q = base_ + prow[mt];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (q < n) f[mt][c] = *reinterpret_cast<const half8_t*>(enc_in + 32 * (size_t)q + c * 8);
                else {
#pragma unroll
                    for (int t = 0; t < 8; t++) f[mt][c][t] = (_Float16)0.f;
                }
            }
        }
    };
    if (MODE == 0) fetch(blockIdx.x * PTS);
    for (int base = blockIdx.x * PTS; base < n; base += gridDim.x * PTS) {
        int pix = 0;
        if (MODE == 1) {   // NT == 2: one point per thread
            const int p = base + threadIdx.x;
            __half enc[32];
            const bool valid = p < n;
            if (valid) { pix = index[p]; float x[3]; normalise_pos(M, pos, pix, x); encode_point(L, M.grid, x, enc); }
            __half* fRow = reinterpret_cast<__half*>(sF) + (size_t)threadIdx.x * MR_FROW;
#pragma unroll
            for (int q = 0; q < 32; q++) fRow[q] = valid ? enc[q] : __float2half(0.f);
            __syncthreads();
        }
        int p[NT];
        f32x16_t acc[NT];
#pragma unroll
        for (int mt = 0; mt < NT; mt++) {
            p[mt] = base + prow[mt];
            if (MODE == 1) {
#pragma unroll
                for (int c = 0; c < 4; c++) f[mt][c] = *reinterpret_cast<const half8_t*>(sF + (size_t)prow[mt] * MR_FROW + c * 8);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][r] = 0.f;
        }
        // ---- layer 0: B = features 2 s + half of the point (fp16 values, exact in fp32)
#pragma unroll
        for (int st = 0; st < 16; st++) {
#pragma unroll
            for (int mt = 0; mt < NT; mt++) {
                const _Float16 e = half ? f[mt][st >> 2][2 * (st & 3) + 1] : f[mt][st >> 2][2 * (st & 3)];
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][st], (float)e, acc[mt], 0, 0, 0);
            }
        }
        if (MODE == 0) fetch(base + gridDim.x * PTS);
        // ---- layers 1, 2: ReLU, halves swapped into (k, k + 1) pairs = next B operand
#pragma unroll
        for (int l = 1; l < 3; l++) {
            float x[NT][16];
#pragma unroll
            for (int mt = 0; mt < NT; mt++) {
#pragma unroll
                for (int r = 0; r < 16; r++) x[mt][r] = __int_as_float(max(__float_as_int(acc[mt][r]), 0));    // ReLU on the bit pattern (integer op: no read hazard games with the MFMA result)
#pragma unroll
                for (int r = 0; r < 16; r += 2) swap_halves(x[mt][r], x[mt][r + 1]);
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mt][r] = 0.f;
            }
#pragma unroll
            for (int st = 0; st < 16; st++) {
#pragma unroll
                for (int mt = 0; mt < NT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[l][st], x[mt][step_reg(st)], acc[mt], 0, 0, 0);
            }
        }
        // ---- epilogue: rows 0..5 are the outputs: lanes 0-31 hold rows 0-3 (regs 0-3), lanes 32-63 rows 4,5 (regs 0,1); sigmoid = the shared
        // fixed arithmetic (include/mirres_fmath.h), the same bits as the per-lane kernel's
#pragma unroll
        for (int mt = 0; mt < NT; mt++) {
            int px = 0;
            if (MODE == 1) px = __shfl(pix, mt * 32 + j, 64);
            float a3 = 0.f;
            if (MODE == 0) a3 = __shfl(acc[mt][3], j, 64);
            const float a[3] = {half ? a3 : acc[mt][0], half ? acc[mt][0] : acc[mt][1], half ? acc[mt][1] : acc[mt][2]};
            if (p[mt] < n) {
                float o[3];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const float lo_ = half ? M.mn[3 + r] : M.mn[r], hi_ = half ? M.mx[3 + r] : M.mx[r];
                    o[r] = mrf_sigmoid(a[r]) * (hi_ - lo_) + lo_;
                }
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 3; r++) out6[6 * (size_t)p[mt] + 3 * half + r] = o[r];
                } else if (half == 0) {
                    float a0 = o[0], a1 = o[1], a2 = o[2];
                    if (use_scale) { a0 = fminf(fmaxf(a0 * sx, 0.f), 1.f); a1 = fminf(fmaxf(a1 * sy, 0.f), 1.f); a2 = fminf(fmaxf(a2 * sz, 0.f), 1.f); }
                    kd[3 * (size_t)px] = a0; kd[3 * (size_t)px + 1] = a1; kd[3 * (size_t)px + 2] = a2;
                } else { rm[2 * (size_t)px] = o[1]; rm[2 * (size_t)px + 1] = o[2]; }
            }
        }
        if (MODE == 1) __syncthreads();
    }
}
// compacted list of pixels whose vertex needs a material lookup (occ >= 0.5): replaces torch.where (renderer_restir.py:398).
// One queue-head word takes ~88 atomics per microsecond, so the list is built with ONE atomic per 4096 slots: a thread looks at 16 consecutive
// slots (four 16-byte loads), the block scans the per-thread counts, and the indices go out in slot order (680 -> ~200 us for 82 M slots; with one
// slot per thread the kernel was nothing but its 80 k atomics).
#define MR_AL_PER 16
__global__ void __launch_bounds__(MR_BLOCK) k_active_list(const float* __restrict__ occ, int n, int32_t* __restrict__ index, uint32_t* __restrict__ count,
                                                          float* __restrict__ kd, int clamp_all) {
    const size_t base = ((size_t)blockIdx.x * MR_BLOCK + threadIdx.x) * MR_AL_PER;
    uint32_t bits = 0;
    if (base + MR_AL_PER <= (size_t)n && (reinterpret_cast<uintptr_t>(occ) & 15) == 0) {   // 16-byte loads only from a 16-byte aligned map (a sliced view may start anywhere)
#pragma unroll
        for (int j = 0; j < MR_AL_PER; j += 4) {
            const float4 o = *reinterpret_cast<const float4*>(occ + base + j);
            bits |= (o.x >= 0.5f ? 1u : 0u) << j | (o.y >= 0.5f ? 2u : 0u) << j | (o.z >= 0.5f ? 4u : 0u) << j | (o.w >= 0.5f ? 8u : 0u) << j;
        }
    } else {
        for (int j = 0; j < MR_AL_PER; j++) if (base + j < (size_t)n && occ[base + j] >= 0.5f) bits |= 1u << j;
    }
    const uint32_t cnt = __popc(bits);
    uint32_t slot = block_append(count, cnt > 0, cnt);
#pragma unroll
    for (int j = 0; j < MR_AL_PER; j++) if (bits & (1u << j)) index[slot++] = (int32_t)(base + j);
    if (clamp_all) {  // torch.clamp(new_diffuse_map, 0, 1) over the whole map when use_scale (:408); active pixels are clamped by the MLP kernel
        for (int j = 0; j < MR_AL_PER; j++) {
            const size_t i = base + j;
            if (i < (size_t)n) {
#pragma unroll
                for (int k = 0; k < 3; k++) kd[3 * i + k] = fminf(fmaxf(kd[3 * i + k], 0.f), 1.f);
            }
        }
    }
}

// the same list from a live-slot list (mirres_render's batches): only the slots whose path is still going are looked at.
// SORT (round 4): every listed slot also gets a Morton key of its position (8 bits per axis of the field's box; 5 in the first version); the list is then sorted by that key
// (k_ls_* below). The indirect vertices of a batch reach the material field in slot
// order (sample-major, pixel-minor): consecutive slots hold hit points of scattered bounce rays, and every point gathers 128 table entries (512 B) of which the
// eleven hashed levels (2 MB each) miss the 4 MB L2 of the XCD. Ordered by position the fused gather + MLP kernel runs 23 / 33 / 36 % faster for 12- / 18- /
// 30-bit keys (scripts/dev_grid_locality.py, profiles/r04_grid_locality.txt); outputs are scattered by slot, so not a bit changes.
#define MR_GS_BITS 8       // default: 24-bit keys, three 8-bit passes; MIRRES_GS_BITS = 1 .. 8 bits per axis (A/B: 5 bits = the two-pass sort of the first version)
MR_DEV uint32_t spread10(uint32_t v) { v &= 0x3ffu; v = (v | (v << 16)) & 0x30000ffu; v = (v | (v << 8)) & 0x300f00fu; v = (v | (v << 4)) & 0x30c30c3u; v = (v | (v << 2)) & 0x9249249u; return v; }
template <bool SORT>
__global__ void __launch_bounds__(MR_BLOCK) k_active_from_live(const float* __restrict__ occ, const int32_t* __restrict__ live, const uint32_t* __restrict__ live_count,
                                                               int32_t* __restrict__ index, uint32_t* __restrict__ count, MatNetD M, const float* __restrict__ pos,
                                                               uint32_t* __restrict__ keys, int bits) {
    const uint32_t nl = *live_count;
    for (uint32_t b0 = blockIdx.x * (MR_BLOCK * 8u); b0 < nl; b0 += gridDim.x * (MR_BLOCK * 8u)) {   // a fixed grid strides over the list
        const uint32_t t0 = b0 + threadIdx.x;   // eight entries per thread (t0 + j * MR_BLOCK: coalesced): one queue atomic per 2048
        int sl[8]; uint32_t n = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            sl[j] = -1;
            if (t0 + j * MR_BLOCK < nl) { const int sv = live[t0 + j * MR_BLOCK]; if (occ[sv] >= 0.5f) { sl[j] = sv; n++; } }
        }
        uint32_t o = block_append(count, n > 0, n);
#pragma unroll
        for (int j = 0; j < 8; j++) if (sl[j] >= 0) {
            if (SORT) {
                uint32_t q[3];
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float u = (pos[3 * (size_t)sl[j] + a] - M.aabb_min[a]) / (M.aabb_max[a] - M.aabb_min[a]);
                    q[a] = (uint32_t)fminf(fmaxf(u * (float)(1 << bits), 0.f), (float)((1 << bits) - 1));      // (NaN -> 0: any bucket will do)
                }
                const uint32_t key = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
                keys[o] = key;
            }
            index[o++] = sl[j];
        }
    }
}
// ---- LSD radix sort (8-bit digits, one pass per key byte) of (key, slot) pairs whose count lives on the device. No global atomics (a first version that counted and
// scattered with one atomic per element cost 2.9 ms per launch for 7.5 M entries, more than the ordered gathers saved): a fixed grid of MR_LS_GRID workgroups,
// each owning a contiguous run of 2048-key tiles — digit histogram of its run -> per digit: exclusive scan over the workgroups -> stable scatter walking the run
// tile by tile with the wave64 digit matching of bvh_build.hip's sort (a key's rank = running base of its digit in this workgroup + the digit's count in the waves
// before + in this wave's earlier rounds + in lower lanes).
#define MR_LS_GRID 1024
#define MR_LS_TILE 2048
MR_DEV void ls_run(uint32_t n, uint32_t& t0, uint32_t& t1) {      // tiles [t0, t1) of this workgroup
    const uint32_t tiles = (n + MR_LS_TILE - 1) / MR_LS_TILE, per = (tiles + MR_LS_GRID - 1) / MR_LS_GRID;
    t0 = blockIdx.x * per; t1 = t0 + per < tiles ? t0 + per : tiles; if (t0 > tiles) t0 = tiles;
}
__global__ void __launch_bounds__(MR_BLOCK) k_ls_hist(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ d_count, int shift, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t n = *d_count; uint32_t t0, t1; ls_run(n, t0, t1);
    for (uint32_t t = t0; t < t1; t++)
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint32_t idx = t * MR_LS_TILE + i * MR_BLOCK + threadIdx.x; if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255u], 1u); }
    __syncthreads();
    hist[(size_t)threadIdx.x * MR_LS_GRID + blockIdx.x] = h[threadIdx.x];      // digit-major
}
// per digit (one workgroup each): exclusive scan over the MR_LS_GRID workgroup counters in place + the digit's total
__global__ void __launch_bounds__(MR_BLOCK) k_ls_scan(uint32_t* __restrict__ hist, uint32_t* __restrict__ totals) {
    __shared__ uint32_t wsum[4];
    uint32_t* h = hist + (size_t)blockIdx.x * MR_LS_GRID;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    uint32_t running = 0u;
    for (int c0 = 0; c0 < MR_LS_GRID; c0 += MR_BLOCK) {
        const int i = c0 + (int)threadIdx.x;
        const uint32_t x = h[i];
        uint32_t inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = 0u, all = 0u;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t t = wsum[w]; if (w < wave) before += t; all += t; }
        h[i] = running + before + inc - x;
        running += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = running;
}
__global__ void __launch_bounds__(MR_BLOCK) k_ls_scatter(const uint32_t* __restrict__ kin, const int32_t* __restrict__ vin, uint32_t* __restrict__ kout, int32_t* __restrict__ vout,
                                                         const uint32_t* __restrict__ d_count, int shift, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ totals) {
    __shared__ uint32_t wh[MR_BLOCK / 64][256];
    __shared__ uint32_t dsum[MR_BLOCK / 64];
    __shared__ uint32_t base[256];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint32_t n = *d_count; uint32_t t0, t1; ls_run(n, t0, t1);
    {   // where this workgroup's keys of digit d (= thread d) start: all keys with smaller digits + this digit's keys in the workgroups before
        const uint32_t tot = totals[threadIdx.x];
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) dsum[wave] = inc;
        __syncthreads();
        uint32_t run = inc - tot + offs[(size_t)threadIdx.x * MR_LS_GRID + blockIdx.x];
        for (int w = 0; w < wave; w++) run += dsum[w];
        base[threadIdx.x] = run;
    }
    for (uint32_t t = t0; t < t1; t++) {
        for (int w = 0; w < MR_BLOCK / 64; w++) wh[w][threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t first = t * MR_LS_TILE + wave * 512u;
        uint32_t k[8], r[8]; int32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t idx = first + i * 64 + lane;
            const bool valid = idx < n;
            k[i] = valid ? kin[idx] : 0xffffffffu; v[i] = valid ? vin[idx] : 0;
            const uint32_t d = (k[i] >> shift) & 255u;
            uint64_t m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) { const bool bit = (d >> b) & 1u; const uint64_t bal = __ballot(bit); m &= bit ? bal : ~bal; }
            const int leader = valid ? __builtin_ctzll(m) : lane;
            uint32_t prev = 0u;
            if (valid && lane == leader) { prev = wh[wave][d]; wh[wave][d] = prev + (uint32_t)__popcll(m); }
            prev = (uint32_t)__shfl((int)prev, leader, 64);
            r[i] = prev + (uint32_t)__popcll(m & lt_mask);
            __syncthreads();
        }
        {   // per digit: the waves' starting places in this tile, and the workgroup's running base moves past the tile
            uint32_t run = base[threadIdx.x];
#pragma unroll
            for (int w = 0; w < MR_BLOCK / 64; w++) { const uint32_t c = wh[w][threadIdx.x]; wh[w][threadIdx.x] = run; run += c; }
            base[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (first + i * 64 + lane < n) { const uint32_t pos = wh[wave][(k[i] >> shift) & 255u] + r[i]; kout[pos] = k[i]; vout[pos] = v[i]; }
        }
        __syncthreads();
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

int launch_matnet_scatter_mfma(const mirres_matnet_t* m, const float* occ, const float* pos, int n, float* kd, float* rm, int use_scale, const float* scale3,
                               int32_t* index, uint32_t* count, hipStream_t s, const int32_t* live, const uint32_t* live_count, const GridSort* gs) {
    float sx = scale3 ? scale3[0] : 1.f, sy = scale3 ? scale3[1] : 1.f, sz = scale3 ? scale3[2] : 1.f;
    MR_HIP(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
    static const bool sort_on = [] { const char* e = getenv("MIRRES_GRID_SORT"); return !(e && e[0] == '0'); }();
    const bool sort = sort_on && live && gs && gs->keys && gs->keys2 && gs->sorted && gs->hist;
    const int32_t* list = index;     // the list the lookup kernel walks
    // (with a live list the whole-map clamp of use_scale is not applied to slots without a vertex: nothing reads their albedo)
    if (live) {
        int ga = grid_for(n, MR_BLOCK * 8); if (ga > 256 * 8) ga = 256 * 8;
        if (sort) {
            static const int bits = [] { const char* e = getenv("MIRRES_GS_BITS"); const int b = e ? atoi(e) : MR_GS_BITS; return b < 1 ? 1 : (b > 8 ? 8 : b); }();
            k_active_from_live<true><<<ga, MR_BLOCK, 0, s>>>(occ, live, live_count, index, count, matd(m), pos, gs->keys, bits);
            uint32_t* const hist = gs->hist; uint32_t* const totals = hist + 256 * MR_LS_GRID;
            // LSD passes over the key's bytes, ping-pong (keys, index) <-> (keys2, sorted); after an even number of passes the sorted list is where the unsorted one was
            uint32_t *ka = gs->keys, *kb = gs->keys2; int32_t *va = index, *vb = gs->sorted;
            for (int shift = 0; shift < 3 * bits; shift += 8) {
                k_ls_hist<<<MR_LS_GRID, MR_BLOCK, 0, s>>>(ka, count, shift, hist);
                k_ls_scan<<<256, MR_BLOCK, 0, s>>>(hist, totals);
                k_ls_scatter<<<MR_LS_GRID, MR_BLOCK, 0, s>>>(ka, va, kb, vb, count, shift, hist, totals);
                uint32_t* tk = ka; ka = kb; kb = tk; int32_t* tv = va; va = vb; vb = tv;
            }
            list = va;
        } else k_active_from_live<false><<<ga, MR_BLOCK, 0, s>>>(occ, live, live_count, index, count, MatNetD(), nullptr, nullptr, 0);
    }
    else k_active_list<<<grid_for(n, MR_BLOCK * MR_AL_PER), MR_BLOCK, 0, s>>>(occ, n, index, count, kd, use_scale);
    int g = grid_for(n, MR_BLOCK); if (g > 256 * 8) g = 256 * 8;
    k_mlp_mfma<1, 2><<<g, MR_BLOCK, 0, s>>>(matd(m), host_levels(nullptr), nullptr, pos, list, count, 0, nullptr, kd, rm, use_scale, sx, sy, sz);
    MR_LAUNCH_CHECK("matnet_scatter_mfma");
    return 0;
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

int mirres_matnet_mlp(const mirres_matnet_t* m, const uint16_t* enc, int n, float* out, void* stream) {
    if (!m || !enc || !out || n < 0) { set_error("mirres_matnet_mlp: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    static const int cap = [] { const char* e = getenv("MIRRES_MLP_GRID"); return e ? atoi(e) : 256 * 3; }();   // 3 resident blocks per CU (~160 registers per lane): every block stages and splits the weights once
    int g = grid_for(n, 256); if (g > cap) g = cap;      // 2 tiles (64 points) per wave
    k_mlp_mfma<0, 2><<<g, MR_BLOCK, 0, (hipStream_t)stream>>>(matd(m), host_levels(nullptr), enc, nullptr, nullptr, nullptr, n, out, nullptr, nullptr, 0, 1.f, 1.f, 1.f);
    MR_LAUNCH_CHECK("matnet_mlp");
    return MIRRES_OK;
}

int mirres_matnet_scatter(const mirres_matnet_t* m, const float* occ, const float* pos, int n, float* kd, float* rough_metal, int use_scale,
                          const float* h_scale3, void* stream) {
    if (!m || !occ || !pos || !kd || !rough_metal || n < 0) { set_error("mirres_matnet_scatter: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    return launch_matnet_scatter(m, occ, pos, n, kd, rough_metal, use_scale, h_scale3, nullptr, nullptr, (hipStream_t)stream);
}

// development aid (not part of include/mirres.h): the PRODUCTION material lookup of mirres_render's indirect vertices — compacted slot list -> fused hash-grid gather + MFMA MLP
// (k_mlp_mfma<1, 2>) -> scatter into kd / rough_metal — on caller-given points, for scripts/dev_grid_locality.py. index: int32[n] scratch, count: uint32[1] scratch.
int mirres_debug_matnet_scatter_mfma(const mirres_matnet_t* m, const float* occ, const float* pos, int n, float* kd, float* rough_metal, int32_t* index, uint32_t* count, void* stream) {
    if (!m || !occ || !pos || !kd || !rough_metal || !index || !count || n <= 0) return MIRRES_E_ARG;
    return launch_matnet_scatter_mfma(m, occ, pos, n, kd, rough_metal, 0, nullptr, index, count, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

}  // extern "C"
