// device_light.hpp — environment light on the device: lookup, 2-D inverse-CDF sampling, density.
// Implements nerf/ScreenSpaceReSTIR/utils/{helper,light,lightDi}.slang behaviour; tables (pdf/cdf/mpdf/mcdf, ~1 MB at
// 256x512) and the 1.5 MB texture stay resident in L2 / Infinity Cache, so lookups are cache traffic, not HBM.
#pragma once
#include "device_math.hpp"

namespace mr {

struct EnvD { const float* tex; int W, H; const float *pdf, *cdf, *mpdf, *mcdf; };

// clamp-to-edge bilinear with int() truncation (helper.slang:46-71) — NOT the wrapping variant of helperDi.slang:80
MR_DEV v3 eval_bi(const float* __restrict__ tex, float u_, float v_, int width, int height) {
    float x = u_ * width - 0.5f, y = v_ * height - 0.5f;
    int x0 = (int)x, y0 = (int)y;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = max(0, min(x0, width - 1)); x1 = max(0, min(x1, width - 1));
    y0 = max(0, min(y0, height - 1)); y1 = max(0, min(y1, height - 1));
    float u = x - x0, v = y - y0;
    v3 t00 = ld3(tex, (size_t)y0 * width + x0), t01 = ld3(tex, (size_t)y0 * width + x1);
    v3 t10 = ld3(tex, (size_t)y1 * width + x0), t11 = ld3(tex, (size_t)y1 * width + x1);
    return (t00 * (1.0f - u) + t01 * u) * (1.0f - v) + (t10 * (1.0f - u) + t11 * u) * v;
}

// env_le (lightDi.slang:119-132)
MR_DEV v3 env_le(v3 dir, const float* __restrict__ tex, int width, int height) {
    float theta = mrf_acos(dir.y);
    float sin_theta = mrf_sin(theta);
    if (fabsf(sin_theta) < 1e-4f) return V3(0.f);
    float phi = mrf_atan2(dir.z, dir.x);
    if (phi < 0) phi += 6.2831853f;
    return eval_bi(tex, phi * 0.1591549f, 1 - theta * 0.31830988f, width, height);
}
// radiance arriving along world direction L (get_light_info, lightDi.slang:285-298)
MR_DEV v3 env_radiance(const EnvD& E, v3 L) { return env_le(ngp_dir(L), E.tex, E.W, E.H); }

// bilinear footprint of env_le(dir) for the backward scatter: 4 texel indices + weights; false at the poles
MR_DEV bool env_le_footprint(v3 dir, int width, int height, int idx[4], float w[4]) {
    float theta = mrf_acos(dir.y);
    float sin_theta = mrf_sin(theta);
    if (fabsf(sin_theta) < 1e-4f) return false;
    float phi = mrf_atan2(dir.z, dir.x);
    if (phi < 0) phi += 6.2831853f;
    float x = (phi * 0.1591549f) * width - 0.5f, y = (1 - theta * 0.31830988f) * height - 0.5f;
    int x0 = (int)x, y0 = (int)y;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = max(0, min(x0, width - 1)); x1 = max(0, min(x1, width - 1));
    y0 = max(0, min(y0, height - 1)); y1 = max(0, min(y1, height - 1));
    float u = x - x0, v = y - y0;
    idx[0] = y0 * width + x0; idx[1] = y0 * width + x1; idx[2] = y1 * width + x0; idx[3] = y1 * width + x1;
    w[0] = (1.0f - u) * (1.0f - v); w[1] = u * (1.0f - v); w[2] = (1.0f - u) * v; w[3] = u * v;
    return true;
}

// upper-bound style search of lightDi.slang:41-52
MR_DEV int find_interval(int left, int right, float val, const float* __restrict__ a) {
    int l = left, r = right;
    while (l < r) {
        int mid = (l + r) / 2;
        if (a[mid] <= val) l = mid + 1; else r = mid;
    }
    return clampi(l - left - 1, 0, right - left);
}

// InfiniteAreaLight_Sample_Li(_no_env): warp_continue + pdf_continue + direction (lightDi.slang:67-105,181-209)
MR_DEV bool sample_li(const EnvD& E, float r0, float r1, v3& dir, float& out_pdf, v2& light_uv) {
    const float PI = 3.141592653589793f;
    float ux = r0, uy = r1;
    const int w_ = E.W, h_ = E.H;
    int row = find_interval(0, h_ + 1, uy, E.mcdf);
    uy = clampf(mr_div(uy - E.mcdf[row], E.mpdf[row]), 0.0f, 1.0f);
    int row_start = row * (w_ + 1);
    int col = find_interval(row_start, row_start + (w_ + 1), ux, E.cdf);
    ux = clampf(mr_div(ux - E.cdf[row * (w_ + 1) + col], E.pdf[row * w_ + col]), 0.0f, 1.0f);
    ux = clampf(mr_div(ux + col, (float)w_), 0.0f, 1.0f);
    uy = clampf(mr_div(uy + row, (float)h_), 0.0f, 1.0f);
    int r2 = clampi(row, 0, h_ - 1), c2 = clampi(col, 0, w_ - 1);
    float pdf = E.pdf[r2 * w_ + c2] * E.mpdf[r2] * w_ * h_;
    float theta = uy * PI, phi = ux * 2 * PI;
    float cos_theta, cos_phi, sin_theta, sin_phi;
    mrf_sincos(theta, &sin_theta, &cos_theta); mrf_sincos(phi, &sin_phi, &cos_phi);
    dir = V3(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi);
    if (fabsf(sin_theta) >= 1e-4f) pdf = mr_div(pdf, 2 * PI * PI * sin_theta);
    else pdf = 0.0f;
    out_pdf = pdf;
    light_uv = V2(ux, 1 - uy);
    return !(pdf == 0);
}

// InfiniteAreaLight_pdf_li (lightDi.slang:312-330)
MR_DEV float pdf_li(const EnvD& E, v3 dir) {
    const float PI = 3.141592653589793f;
    v3 w = V3(clampf(dir.x, -1.0f, 1.0f), clampf(dir.y, -1.0f, 1.0f), clampf(dir.z, -1.0f, 1.0f));
    float theta = mrf_acos(w.y);
    float sin_theta = mrf_sin(theta);
    if (fabsf(sin_theta) < 1e-4f) return 0;
    float phi = mrf_atan2(w.z, w.x);
    if (phi < 0) phi += 6.2831853f;
    int col = (int)(phi * 0.1591549f * E.W);
    int row = (int)(theta * 0.31830988f * E.H);
    row = clampi(row, 0, E.H - 1); col = clampi(col, 0, E.W - 1);
    return mr_div(E.pdf[row * E.W + col] * E.mpdf[row] * E.W * E.H, 2 * PI * PI * sin_theta);
}

// uv2xy (helper.slang:26-36)
MR_DEV void uv2xy(v2 uv, int width, int height, int& ox, int& oy) {
    float x = uv.x * width, y = uv.y * height;
    int x0 = x < 0.f ? (int)x - 1 : (int)x;
    int y0 = y < 0.f ? (int)y - 1 : (int)y;
    ox = ((x0 % width) + width) % width;
    oy = ((y0 % height) + height) % height;
}

}  // namespace mr
