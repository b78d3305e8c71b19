// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header). PARITY UNPINNED.
// Material field = tiny-cuda-nn HashGrid encoding + bias-free 32-32-32-6 ReLU MLP + sigmoid/affine
// (nerf/render_helper.py:28-117). tiny-cuda-nn is an un-vendored, un-pinned third-party dependency
// (readme.md:30); its published algorithm (include/tiny-cuda-nn/encodings/grid.h, as known at survey time)
// is restated here: fp16 parameters, fp16 accumulation of the 8-corner interpolation, pos = fmaf(scale,x,0.5).
#pragma once
#include "orc_math.hpp"
#include <vector>

namespace orc {

struct GridLevel { float scale; uint32_t resolution; uint32_t size; uint32_t offset; };  // offset/size in entries (x2 features)

struct HashGridCfg {
    int n_levels = 16, n_features = 2, log2_hashmap_size = 19, base_resolution = 16;
    float per_level_scale = 1.4472692012786865f;  // exp(log(4096/16)/15) rounded to fp32
};

static inline std::vector<GridLevel> grid_levels(const HashGridCfg& c, uint32_t* total_entries) {
    std::vector<GridLevel> L(c.n_levels);
    uint32_t offset = 0;
    float log2_pls = log2f(c.per_level_scale);
    for (int i = 0; i < c.n_levels; i++) {
        float scale = exp2f(i * log2_pls) * c.base_resolution - 1.0f;      // grid_scale
        uint32_t res = (uint32_t)ceilf(scale) + 1;                        // grid_resolution
        uint64_t dense = (uint64_t)res * res * res;
        uint32_t params = dense > 0x7fffffffull ? 0x7fffffffu : (uint32_t)dense;
        params = (params + 7u) / 8u * 8u;                                 // next_multiple(.,8)
        params = std::min(params, 1u << c.log2_hashmap_size);
        L[i].scale = scale; L[i].resolution = res; L[i].size = params; L[i].offset = offset;
        offset += params;
    }
    if (total_entries) *total_entries = offset;
    return L;
}

static inline uint32_t grid_index(const GridLevel& l, uint32_t px, uint32_t py, uint32_t pz) {
    uint32_t stride = 1, index = 0;
    const uint32_t p[3] = {px, py, pz};
    for (int d = 0; d < 3; d++) {
        if (stride <= l.size) { index += p[d] * stride; stride *= l.resolution; }
    }
    if (l.size < stride) index = (px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u);
    return index % l.size;
}

// encode one point x in [0,1]^3 -> 32 halfs (level-major). params: fp16 table [total_entries*2].
static inline void hashgrid_encode(const std::vector<GridLevel>& L, const uint16_t* params, const float x[3], uint16_t* out) {
    for (size_t lv = 0; lv < L.size(); lv++) {
        const GridLevel& l = L[lv];
        float pos[3]; uint32_t pg[3];
        for (int d = 0; d < 3; d++) {
            float p = fmaf(l.scale, x[d], 0.5f);
            float fl = floorf(p);
            pg[d] = (uint32_t)(int)fl;
            pos[d] = p - fl;
        }
        uint16_t r0 = 0, r1 = 0;
        for (uint32_t idx = 0; idx < 8; idx++) {
            float w = 1.f; uint32_t pl[3];
            for (int d = 0; d < 3; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                else { w *= pos[d]; pl[d] = pg[d] + 1; }
            }
            uint32_t gi = grid_index(l, pl[0], pl[1], pl[2]);
            const uint16_t* v = params + 2 * ((size_t)l.offset + gi);
            r0 = f16_add(r0, f32_to_f16(w * f16_to_f32(v[0])));
            r1 = f16_add(r1, f32_to_f16(w * f16_to_f32(v[1])));
        }
        out[2 * lv] = r0; out[2 * lv + 1] = r1;
    }
}

struct MatNet {
    std::vector<GridLevel> levels; const uint16_t* params;
    const float* w0; const float* w1; const float* w2;  // torch Linear weights [out,in]: [32,32],[32,32],[6,32]
    float aabb_min[3], aabb_max[3]; float mn[6], mx[6];
};

// MLPTexture3D.sample_no_di  render_helper.py:106-117. Linear layers are k-ordered fmaf chains (fp32).
static inline void matnet_eval(const MatNet& M, const float p[3], float out[6]) {
    float x[3];
    for (int d = 0; d < 3; d++) x[d] = fminf(fmaxf((p[d] - M.aabb_min[d]) / (M.aabb_max[d] - M.aabb_min[d]), 0.f), 1.f);
    uint16_t enc[32];
    hashgrid_encode(M.levels, M.params, x, enc);
    float a[32], h[32];
    for (int i = 0; i < 32; i++) a[i] = f16_to_f32(enc[i]);
    for (int o = 0; o < 32; o++) { float acc = 0.f; for (int k = 0; k < 32; k++) acc = fmaf(a[k], M.w0[o * 32 + k], acc); h[o] = fmaxf(acc, 0.f); }
    for (int o = 0; o < 32; o++) { float acc = 0.f; for (int k = 0; k < 32; k++) acc = fmaf(h[k], M.w1[o * 32 + k], acc); a[o] = fmaxf(acc, 0.f); }
    for (int o = 0; o < 6; o++) {
        float acc = 0.f; for (int k = 0; k < 32; k++) acc = fmaf(a[k], M.w2[o * 32 + k], acc);
        float s = mrf_sigmoid(acc);
        out[o] = s * (M.mx[o] - M.mn[o]) + M.mn[o];
    }
}

}  // namespace orc
