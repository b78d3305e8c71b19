// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
// Restatement of the reference's non-ReSTIR direct-lighting renderer, nerf/render_dump.py (BASELINE configs[0]: --use_brdf without
// --use_restir, nerf/renderer.py:1131-1149). Unlike the Slang path this one IS pinned: render_dump.py is pure torch and is run as-is
// (imported from /root/reference, CPU tensors, a brute-force numpy Moeller-Trumbore loop as the `intersector`) by tests/golden/gen_from_reference.py;
// tests/test_oracle_golden.py holds this file to that fixture.
#pragma once
#include "orc_kernels.hpp"

namespace orc {

// F.normalize(x, p=2, dim=-1, eps): x / max(||x||, eps)   (render_dump.py:5-6, :39-42, :161)
static inline f3 torch_normalize(f3 v, float eps) {
    const float n = sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
    const float d = fmaxf(n, eps);
    return mk3(v.x / d, v.y / d, v.z / d);
}

// The `intersector` render_dump.py takes from outside is a conventional ray tracer: "is some triangle hit in front of the origin". Same
// hierarchy and box test as bvh_hit, but a leaf counts only with t > 0 and nothing shrinks the interval (an occlusion query has no order).
static inline bool bvh_occluded_front(const Bvh& B, f3 o, f3 d, float t_min, float t_max) {
    d = normalize(d);
    int stack[128]; int count = 0;
    stack[count++] = 0;
    while (count > 0) {
        const int idx = stack[--count];
        if (!aabb_hit(o, d, t_min, t_max, B.aabb + 6 * (size_t)idx)) continue;
        const int left = B.info[3 * idx], right = B.info[3 * idx + 1];
        if (left != 0 && right != 0) { if (count + 2 <= 128) { stack[count++] = left; stack[count++] = right; } }
        else if (left == 0 && right == 0) {
            const int32_t* ti = B.tri + 3 * (size_t)B.info[3 * idx + 2];
            float t = 0.f; f3 nn = mk3(1.f);
            if (triangle_hit(o, d, ld3(B.vert, ti[0]), ld3(B.vert, ti[1]), ld3(B.vert, ti[2]), t, false, nn) && t > 0.f) return true;
        }
    }
    return false;
}

// get_light_rgbs (render_dump.py:70-82): lat-long lookup through F.grid_sample(bilinear, zeros padding, align_corners=False)
static inline f3 dump_light_rgb(const float* env /*[H,W,3]*/, int H, int W, f3 d) {
    const float PI = 3.14159265358979323846f;
    const float phi = mrf_acos(d.z) - 1e-6f;
    const float theta = mrf_atan2(d.y, d.x);
    const float qy = (phi / PI) * 2 - 1;
    const float qx = -theta / PI;
    const float x = ((qx + 1) * W - 1) / 2, y = ((qy + 1) * H - 1) / 2;   // grid_sampler_unnormalize, align_corners=False
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = x - x0f, wx0 = (x0f + 1) - x, wy1 = y - y0f, wy0 = (y0f + 1) - y;   // nw = (ix_se - ix) * (iy_se - iy) ...
    f3 r = mk3(0.f);
    auto tap = [&](int xi, int yi, float w) { if (xi >= 0 && xi < W && yi >= 0 && yi < H) { const float* p = env + 3 * ((size_t)yi * W + xi); r = r + mk3(p[0] * w, p[1] * w, p[2] * w); } };
    tap(x0, y0, wx0 * wy0); tap(x1, y0, wx1 * wy0); tap(x0, y1, wx0 * wy1); tap(x1, y1, wx1 * wy1);
    return r;
}

// GGX_specular (render_dump.py:32-65) for one (surface point, light) pair; roughness / fresnel are 3-channel maps (renderer.py:1134-1135)
static inline f3 ggx_specular(f3 normal, f3 pts2c, f3 pts2l, f3 rough, f3 fresnel) {
    const float PI = 3.14159265358979323846f;
    const f3 L = torch_normalize(pts2l, 1e-12f), V = torch_normalize(pts2c, 1e-12f);
    const f3 Hh = torch_normalize(mk3((L.x + V.x) / 2.0f, (L.y + V.y) / 2.0f, (L.z + V.z) / 2.0f), 1e-12f);
    f3 N = torch_normalize(normal, 1e-12f);
    const float nov0 = (V.x * N.x + V.y * N.y) + V.z * N.z;
    const float sg = nov0 > 0.f ? 1.f : (nov0 < 0.f ? -1.f : 0.f);
    N = N * sg;
    const float NoL = clampf((N.x * L.x + N.y * L.y) + N.z * L.z, 1e-6f, 1.f);
    const float NoV = clampf((N.x * V.x + N.y * V.y) + N.z * V.z, 1e-6f, 1.f);
    const float NoH = clampf((N.x * Hh.x + N.y * Hh.y) + N.z * Hh.z, 1e-6f, 1.f);
    const float VoH = clampf((V.x * Hh.x + V.y * Hh.y) + V.z * Hh.z, 1e-6f, 1.f);
    const float FMi = ((-5.55473f) * VoH - 6.98316f) * VoH;
    const float p2 = mrf_exp2(FMi);
    float out[3]; const float r3[3] = {rough.x, rough.y, rough.z}, f3_[3] = {fresnel.x, fresnel.y, fresnel.z};
    for (int c = 0; c < 3; c++) {
        const float alpha = r3[c] * r3[c], alpha2 = alpha * alpha;
        const float k = (alpha + 2 * r3[c] + 1.0f) / 8.0f;
        const float frac0 = f3_[c] + (1 - f3_[c]) * p2;
        const float frac = frac0 * alpha2;
        const float nom0 = NoH * NoH * (alpha2 - 1) + 1;
        const float nom1 = NoV * (1 - k) + k;
        const float nom2 = NoL * (1 - k) + k;
        const float nom = clampf(4 * PI * nom0 * nom0 * nom1 * nom2, 1e-6f, 4 * PI);
        out[c] = frac / nom;
    }
    return mk3(out[0], out[1], out[2]);
}

// dump_render_run_mesh (render_dump.py:136-215) for one surface point: sums over the fixed light set. `equal_areas` selects the
// 'stratifed_sample_equal_areas' branch (mean of 4 pi f L cos) instead of the area-weighted sum. Visibility: batch_intersector (:8-27) —
// origin pos + d * 0.001, a hit in front of it (bvh_occluded_front) zeroes the light; lights with cosine <= 1e-6 are not traced (visibility stays 1).
static inline void dump_render_point(const Bvh& B, f3 pos, f3 normal, f3 albedo, f3 rough, f3 fresnel, f3 ray_d, int L, const float* light_dirs,
                                     const float* light_w, const float* light_rgb, bool equal_areas, f3& rgb, f3& diff, f3& spec) {
    const float PI = 3.14159265358979323846f;
    const f3 surf2c = torch_normalize(-ray_d, 1e-6f);
    f3 s_all = mk3(0.f), s_d = mk3(0.f), s_s = mk3(0.f);
    for (int l = 0; l < L; l++) {
        const f3 d = ld3(light_dirs, l);
        float cosine = (d.x * normal.x + d.y * normal.y) + d.z * normal.z;
        cosine = fmaxf(cosine, 0.0f);
        float vis = 1.f;
        if (cosine > 1e-6f) {
            if (bvh_occluded_front(B, pos + d * 0.001f, d, 0.f, 1e7f)) vis = 0.f;
        }
        const f3 sp = ggx_specular(normal, surf2c, d, rough, fresnel);
        const f3 bd = mk3(albedo.x / PI, albedo.y / PI, albedo.z / PI);
        const f3 light = ld3(light_rgb, l) * vis;
        const float dconst = (float)(1.0 / 3.14159265358979323846);   // surface_brdf_diff = 1/np.pi, a Python float
        if (equal_areas) {
            s_d = s_d + mk3(4 * PI * dconst * light.x * cosine, 4 * PI * dconst * light.y * cosine, 4 * PI * dconst * light.z * cosine);
            s_s = s_s + mk3(4 * PI * sp.x * light.x * cosine, 4 * PI * sp.y * light.y * cosine, 4 * PI * sp.z * light.z * cosine);
            s_all = s_all + mk3(4 * PI * (bd.x + sp.x) * light.x * cosine, 4 * PI * (bd.y + sp.y) * light.y * cosine, 4 * PI * (bd.z + sp.z) * light.z * cosine);
        } else {
            const float w = light_w[l];
            s_d = s_d + mk3(dconst * light.x * cosine * w, dconst * light.y * cosine * w, dconst * light.z * cosine * w);
            s_s = s_s + mk3(sp.x * light.x * cosine * w, sp.y * light.y * cosine * w, sp.z * light.z * cosine * w);
            s_all = s_all + mk3((bd.x + sp.x) * light.x * cosine * w, (bd.y + sp.y) * light.y * cosine * w, (bd.z + sp.z) * light.z * cosine * w);
        }
    }
    if (equal_areas) { const float inv = (float)L; s_d = s_d / inv; s_s = s_s / inv; s_all = s_all / inv; }
    rgb = s_all; diff = s_d; spec = s_s;
}

}  // namespace orc
