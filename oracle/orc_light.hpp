// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header). PARITY UNPINNED.
// Environment light: lookup, importance tables, sampling. Restated from
// nerf/ScreenSpaceReSTIR/utils/{helper,light,lightDi}.slang, make_sampleable.slang, GenerateLightTiles.py.
#pragma once
#include "orc_math.hpp"
#include <vector>

namespace orc {

struct Env {
    const float* tex;   // [Hc*Wc,3], already vertically flipped (renderer_restir.py:305-311)
    int W, H;
    const float* pdf;   // [Hc*Wc]
    const float* cdf;   // [Hc*(Wc+1)]
    const float* mpdf;  // [Hc]
    const float* mcdf;  // [Hc+1]
};

// utils/helper.slang:46-71  (clamp-to-edge bilinear with int() truncation)
static inline f3 eval_bi(const float* tex, f2 uv, int width, int height) {
    float x = uv.x * width - 0.5f;
    float y = uv.y * height - 0.5f;
    int x0 = (int)x, y0 = (int)y;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = std::max(0, std::min(x0, width - 1));
    x1 = std::max(0, std::min(x1, width - 1));
    y0 = std::max(0, std::min(y0, height - 1));
    y1 = std::max(0, std::min(y1, height - 1));
    float u = x - x0, v = y - y0;
    const float* p00 = tex + 3 * ((size_t)y0 * width + x0);
    const float* p01 = tex + 3 * ((size_t)y0 * width + x1);
    const float* p10 = tex + 3 * ((size_t)y1 * width + x0);
    const float* p11 = tex + 3 * ((size_t)y1 * width + x1);
    f3 t00 = mk3(p00[0], p00[1], p00[2]), t01 = mk3(p01[0], p01[1], p01[2]);
    f3 t10 = mk3(p10[0], p10[1], p10[2]), t11 = mk3(p11[0], p11[1], p11[2]);
    // math_lerp  helper.slang:3-7
    return (t00 * (1.0f - u) + t01 * u) * (1.0f - v) + (t10 * (1.0f - u) + t11 * u) * v;
}

// utils/lightDi.slang:119-132
static inline f3 env_le(f3 dir, const float* tex, int width, int height) {
    const float TWO_PI = 6.2831853f, INV_TWO_PI = 0.1591549f, INV_PI = 0.31830988f;
    float theta = mrf_acos(dir.y);
    float sin_theta = mrf_sin(theta);
    if (fabsf(sin_theta) < 1e-4f) return mk3(0.f);
    float phi = mrf_atan2(dir.z, dir.x);
    if (phi < 0) phi += TWO_PI;
    f2 uv = mk2(phi * INV_TWO_PI, 1 - theta * INV_PI);
    return eval_bi(tex, uv, width, height);
}

// lightDi.slang:285-298 get_light_info
static inline void get_light_info(const Env& E, f2 light_uv, f3& emission, f3& dir) {
    dir = oct_decode(light_uv);
    emission = env_le(ngp_dir(dir), E.tex, E.W, E.H);
}

// helper.slang:26-36 uv2xy
static inline void uv2xy(f2 uv, int width, int height, int& ox, int& oy) {
    float x = uv.x * width, y = uv.y * height;
    int x0 = x < 0.f ? (int)x - 1 : (int)x;
    int y0 = y < 0.f ? (int)y - 1 : (int)y;
    ox = ((x0 % width) + width) % width;
    oy = ((y0 % height) + height) % height;
}

// lightDi.slang:41-52
static inline int find_interval(int left, int right, float val, const float* a) {
    int l = left, r = right;
    while (l < r) {
        int mid = (l + r) / 2;
        if (a[mid] <= val) l = mid + 1; else r = mid;
    }
    return clampi(l - left - 1, 0, right - left);
}

// warp_continue + pdf_continue + direction (lightDi.slang:67-105, 181-209; light.slang:105-137)
// returns false when pdf == 0. light_uv = (uv.x, 1-uv.y) continuous.
static inline bool sample_li(const Env& E, f2 rnd, f3& dir, float& out_pdf, f2& light_uv) {
    const float PI = 3.141592653589793f;
    f2 uv = rnd;
    int w_ = E.W, h_ = E.H;
    int row = find_interval(0, h_ + 1, uv.y, E.mcdf);
    uv.y = clampf((uv.y - E.mcdf[row]) / E.mpdf[row], 0.0f, 1.0f);
    int row_start = row * (w_ + 1), row_end = row_start + (w_ + 1);
    int col = find_interval(row_start, row_end, uv.x, E.cdf);
    int ic = row * (w_ + 1) + col, ip = row * w_ + col;
    uv.x = clampf((uv.x - E.cdf[ic]) / E.pdf[ip], 0.0f, 1.0f);
    uv.x = clampf((uv.x + col) / w_, 0.0f, 1.0f);
    uv.y = clampf((uv.y + row) / h_, 0.0f, 1.0f);
    int r2 = clampi(row, 0, h_ - 1), c2 = clampi(col, 0, w_ - 1);
    float pdf = E.pdf[r2 * w_ + c2] * E.mpdf[r2] * w_ * h_;
    float theta = uv.y * PI, phi = uv.x * 2 * PI;
    float cos_theta = mrf_cos(theta), cos_phi = mrf_cos(phi), sin_theta = mrf_sin(theta), sin_phi = mrf_sin(phi);
    dir = mk3(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi);
    if (fabsf(sin_theta) >= 1e-4f) pdf = pdf / (2 * PI * PI * sin_theta);
    else pdf = 0.0f;
    out_pdf = pdf;
    light_uv = mk2(uv.x, 1 - uv.y);
    return !(pdf == 0);
}

// lightDi.slang:312-330 InfiniteAreaLight_pdf_li
static inline float pdf_li(const Env& E, f3 dir) {
    const float TWO_PI = 6.2831853f, INV_TWO_PI = 0.1591549f, INV_PI = 0.31830988f, PI = 3.141592653589793f;
    f3 w = mk3(clampf(dir.x, -1.0f, 1.0f), clampf(dir.y, -1.0f, 1.0f), clampf(dir.z, -1.0f, 1.0f));
    float theta = mrf_acos(w.y);
    float sin_theta = mrf_sin(theta);
    if (fabsf(sin_theta) < 1e-4f) return 0;
    float phi = mrf_atan2(w.z, w.x);
    if (phi < 0) phi += TWO_PI;
    int col = (int)(phi * INV_TWO_PI * E.W);
    int row = (int)(theta * INV_PI * E.H);
    row = clampi(row, 0, E.H - 1); col = clampi(col, 0, E.W - 1);
    return (E.pdf[row * E.W + col] * E.mpdf[row] * E.W * E.H) / (2 * PI * PI * sin_theta);
}

// kernel `make_sampleable` alone (make_sampleable.slang:34-60): un-normalised texel weights
static inline void env_weights(const float* tex, int W, int H, float* weight) {
    const float PI = 3.141592653589793f;
    for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++) {
            float v = (h + .5f) / H;
            float sin_theta = mrf_sin(PI * v);
            f2 uv = mk2((w + .5f) / W, v);
            float theta = uv.y * PI, phi = uv.x * 2 * PI;
            float cos_theta = mrf_cos(theta), cos_phi = mrf_cos(phi), sin_theta_dir = mrf_sin(theta), sin_phi = mrf_sin(phi);
            f3 raw = mk3(sin_theta_dir * cos_phi, cos_theta, sin_theta_dir * sin_phi);
            float wv = luminance(env_le(ngp_dir(raw), tex, W, H));
            wv *= sin_theta;
            weight[h * W + w] = wv;
        }
}
// kernel `Distribution2D` alone (make_sampleable.slang:62-86), in place
static inline void distribution2d(int W, int H, float* pdf, float* cdf) {
    for (int y = 0; y < H; y++) {
        float row_weight = cdf[y * (W + 1) + W];
        for (int x = 0; x < W; x++) {
            if (row_weight < 1e-4f) { pdf[y * W + x] = 1.0f / W; cdf[y * (W + 1) + x] = x / (float)W; }
            else { pdf[y * W + x] /= row_weight; cdf[y * (W + 1) + x] /= row_weight; }
        }
    }
}

// make_sampleable: kernels make_sampleable.slang:34-86 + torch glue GenerateLightTiles.py:4-29.
// Sequential fp32 sums stand in for torch's cumsum/sum (summation order differs by ulps on a GPU).
static inline void make_sampleable(const float* tex, int W, int H, float* pdf, float* cdf, float* mpdf, float* mcdf) {
    const float PI = 3.141592653589793f;
    for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++) {
            float v = (h + .5f) / H;
            float sin_theta = mrf_sin(PI * v);
            f2 uv = mk2((w + .5f) / W, v);
            float theta = uv.y * PI, phi = uv.x * 2 * PI;
            float cos_theta = mrf_cos(theta), cos_phi = mrf_cos(phi), sin_theta_dir = mrf_sin(theta), sin_phi = mrf_sin(phi);
            f3 raw = mk3(sin_theta_dir * cos_phi, cos_theta, sin_theta_dir * sin_phi);
            float wv = luminance(env_le(ngp_dir(raw), tex, W, H));
            wv *= sin_theta;
            pdf[h * W + w] = wv;
        }
    mcdf[0] = 0.f;
    float macc = 0.f;
    for (int h = 0; h < H; h++) {
        float acc = 0.f;
        cdf[h * (W + 1)] = 0.f;
        for (int w = 0; w < W; w++) { acc += pdf[h * W + w]; cdf[h * (W + 1) + w + 1] = acc; }
        mpdf[h] = acc;            // pdf_.sum(1)
        macc += acc; mcdf[h + 1] = macc;  // mpdf_.cumsum(0)
    }
    // Distribution2D  make_sampleable.slang:62-86
    for (int y = 0; y < H; y++) {
        float row_weight = cdf[y * (W + 1) + W];
        for (int x = 0; x < W; x++) {
            if (row_weight < 1e-4f) { pdf[y * W + x] = 1.0f / W; cdf[y * (W + 1) + x] = x / (float)W; }
            else { pdf[y * W + x] /= row_weight; cdf[y * (W + 1) + x] /= row_weight; }
        }
        cdf[y * (W + 1) + W] = 1.f;
    }
    float total = mcdf[H];
    for (int h = 0; h < H; h++) mpdf[h] = mpdf[h] / total;
    for (int h = 0; h <= H; h++) mcdf[h] = mcdf[h] / total;
    mcdf[H] = 1.f;
}

// createNeighborOffsetTexture  make_sampleable.slang:186-205 (raw = what the kernel writes); load_m_for_restir then divides by 127
// (renderer_restir.py:220-221)
static inline void neighbor_offsets_raw(int count, float* out) {
    const int R = 254;
    const float phi2 = 1.f / 1.3247179572447f;
    float u = 0.5f, v = 0.5f;
    for (uint32_t index = 0; index < (uint32_t)count * 2;) {
        u += phi2; v += phi2 * phi2;
        if (u >= 1.f) u -= 1.f;
        if (v >= 1.f) v -= 1.f;
        float rSq = (u - 0.5f) * (u - 0.5f) + (v - 0.5f) * (v - 0.5f);
        if (rSq > 0.25f) continue;
        out[index++] = (float)(int)((u - 0.5f) * R);
        out[index++] = (float)(int)((v - 0.5f) * R);
    }
}
static inline void neighbor_offsets(int count, float* out) {
    neighbor_offsets_raw(count, out);
    for (int i = 0; i < 2 * count; i++) out[i] = out[i] / 127;
}

// process_GenerateLightTiles  GenerateLightTiles.slang:16-62 (scalar seeds splat to both lanes)
static inline void light_tiles(const Env& E, uint32_t frameIndex, int tile_count, int tile_size,
                               float* light_data, int32_t* light_uv, float* light_pdf) {
    for (int tile = 0; tile < tile_count; tile++)
        for (int s = 0; s < tile_size; s++) {
            uint32_t idx = (uint32_t)tile * tile_size + s;
            uint32_t sg = seed_generator(idx, idx, frameIndex + 1);
            f3 ld = mk3(0.f); int ux = 0, uy = 0; float ip = 0.f;
            float r0 = next1d(sg), r1 = next1d(sg);  // float2(sampleNext1D, sampleNext1D): source order
            f3 dir; float pdf; f2 luv;
            if (sample_li(E, mk2(r0, r1), dir, pdf, luv)) {
                f2 o = oct_encode(dir);
                ld = mk3(1.f, o.x, o.y);
                uv2xy(luv, E.W, E.H, ux, uy);
                ip = pdf;
            }
            light_data[3 * idx] = ld.x; light_data[3 * idx + 1] = ld.y; light_data[3 * idx + 2] = ld.z;
            light_uv[2 * idx] = ux; light_uv[2 * idx + 1] = uy;
            light_pdf[idx] = ip;
        }
}

}  // namespace orc
