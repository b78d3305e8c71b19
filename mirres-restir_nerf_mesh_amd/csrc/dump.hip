// dump.hip — the reference's non-ReSTIR direct-lighting renderer (nerf/render_dump.py; `--use_brdf` without `--use_restir`,
// nerf/renderer.py:1131-1149; BASELINE configs[0]): every surface point is lit by a FIXED lat-long light set (generate_envir_map_dir,
// nerf/render_helper.py:8-26), each light gated by one occlusion ray, shaded with TensoIR's GGX (GGX_specular, render_dump.py:32-65).
// The reference materialises [points, lights, 3] tensors per chunk; here the pairs above the horizon become shadow rays of one queue
// (generate -> trace -> shade, the same wavefront split as the ReSTIR passes) and a thread per point sums its lights in index order.
// The occlusion query is the conventional one (hit in front of the origin): the reference takes its `intersector` from outside
// (nerf/renderer.py:179), it is not the ReSTIR path's bvh_hit with its t-blind triangle test.
#include "engine.hpp"
#include "device_math.hpp"

namespace mr {

#define MR_BLOCK 256
#define MR_DUMP_BLOCK 1024

// F.normalize(x, p=2, dim=-1, eps): x / max(||x||, eps)
MR_DEV v3 torch_normalize(v3 v, float eps) {
    const float n = sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
    const float d = fmaxf(n, eps);
    return V3(v.x / d, v.y / d, v.z / d);
}

// get_light_rgbs (render_dump.py:70-82): F.grid_sample(bilinear, zeros padding, align_corners=False) of the [H,W,3] map
__global__ void __launch_bounds__(MR_BLOCK) k_dump_light_rgb(const float* __restrict__ env, int H, int W, const float* __restrict__ dirs, int L, float* __restrict__ out) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const float PI = 3.14159265358979323846f;
    const v3 d = ld3(dirs, l);
    const float phi = mrf_acos(d.z) - 1e-6f;
    const float theta = mrf_atan2(d.y, d.x);
    const float qy = (phi / PI) * 2 - 1;
    const float qx = -theta / PI;
    const float x = ((qx + 1) * W - 1) / 2, y = ((qy + 1) * H - 1) / 2;
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = x - x0f, wx0 = (x0f + 1) - x, wy1 = y - y0f, wy0 = (y0f + 1) - y;
    v3 r = V3(0.f);
    const int xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
    const float ws[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (xs[k] >= 0 && xs[k] < W && ys[k] >= 0 && ys[k] < H) { const v3 t = ld3(env, (size_t)ys[k] * W + xs[k]); r = r + V3(t.x * ws[k], t.y * ws[k], t.z * ws[k]); }
    }
    st3(out, l, r);
}

// batch_intersector (render_dump.py:8-27) input: one occlusion ray per (point, light) pair whose clamped cosine exceeds 1e-6 (:166-174)
__global__ void __launch_bounds__(MR_DUMP_BLOCK) k_dump_gen(int p0, int np, int L, const float* __restrict__ pos, const float* __restrict__ normal,
                                                            const float* __restrict__ dirs, Ray* __restrict__ q, uint32_t* __restrict__ q_count, int32_t* __restrict__ slot_out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)np * (size_t)L;
    bool want = false; v3 o = V3(0.f), d = V3(0.f);
    if (idx < total) {
        const int l = (int)(idx / (size_t)np), p = p0 + (int)(idx % (size_t)np);   // light-major: neighbouring rays share the direction and start at neighbouring points
        d = ld3(dirs, l);
        const v3 n = ld3(normal, p);
        const float cosine = fmaxf((d.x * n.x + d.y * n.y) + d.z * n.z, 0.0f);
        if (cosine > 1e-6f) { want = true; o = ld3(pos, p) + d * 0.001f; }   // rays_o + rays_d * vis_near (:12)
    }
    const uint32_t slot = block_append(q_count, want);
    if (want) {
        float4 a, b; a.x = o.x; a.y = o.y; a.z = o.z; a.w = 0.f; b.x = d.x; b.y = d.y; b.z = d.z; b.w = 1e7f;
        reinterpret_cast<float4*>(q + slot)[0] = a; reinterpret_cast<float4*>(q + slot)[1] = b;
    }
    if (idx < total) slot_out[idx] = want ? (int32_t)slot : -1;
}

// GGX_specular (render_dump.py:32-65) for one (point, light) pair
MR_DEV v3 ggx_specular(v3 normal, v3 pts2c, v3 pts2l, v3 rough, v3 fresnel) {
    const float PI = 3.14159265358979323846f;
    const v3 Ld = torch_normalize(pts2l, 1e-12f), Vd = torch_normalize(pts2c, 1e-12f);
    const v3 Hh = torch_normalize(V3((Ld.x + Vd.x) / 2.0f, (Ld.y + Vd.y) / 2.0f, (Ld.z + Vd.z) / 2.0f), 1e-12f);
    v3 N = torch_normalize(normal, 1e-12f);
    const float nov0 = (Vd.x * N.x + Vd.y * N.y) + Vd.z * N.z;
    const float sg = nov0 > 0.f ? 1.f : (nov0 < 0.f ? -1.f : 0.f);
    N = N * sg;
    const float NoL = fminf(fmaxf((N.x * Ld.x + N.y * Ld.y) + N.z * Ld.z, 1e-6f), 1.f);
    const float NoV = fminf(fmaxf((N.x * Vd.x + N.y * Vd.y) + N.z * Vd.z, 1e-6f), 1.f);
    const float NoH = fminf(fmaxf((N.x * Hh.x + N.y * Hh.y) + N.z * Hh.z, 1e-6f), 1.f);
    const float VoH = fminf(fmaxf((Vd.x * Hh.x + Vd.y * Hh.y) + Vd.z * Hh.z, 1e-6f), 1.f);
    const float FMi = ((-5.55473f) * VoH - 6.98316f) * VoH;
    const float p2 = mrf_exp2(FMi);
    const float r3[3] = {rough.x, rough.y, rough.z}, f3[3] = {fresnel.x, fresnel.y, fresnel.z};
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float alpha = r3[c] * r3[c], alpha2 = alpha * alpha;
        const float k = (alpha + 2 * r3[c] + 1.0f) / 8.0f;
        const float frac = (f3[c] + (1 - f3[c]) * p2) * alpha2;
        const float nom0 = NoH * NoH * (alpha2 - 1) + 1;
        const float nom1 = NoV * (1 - k) + k;
        const float nom2 = NoL * (1 - k) + k;
        const float nom = fminf(fmaxf(4 * PI * nom0 * nom0 * nom1 * nom2, 1e-6f), 4 * PI);
        out[c] = frac / nom;
    }
    return V3(out[0], out[1], out[2]);
}

// dump_render_run_mesh (render_dump.py:136-215) + the clamp of dump_render (:129): one thread per surface point, lights in index order
__global__ void __launch_bounds__(MR_BLOCK) k_dump_shade(int p0, int np, int L, const float* __restrict__ normal, const float* __restrict__ albedo,
                                                         const float* __restrict__ rough, const float* __restrict__ fresnel, const float* __restrict__ rays_d,
                                                         const float* __restrict__ dirs, const float* __restrict__ light_w, const float* __restrict__ light_rgb,
                                                         const int32_t* __restrict__ slot, const int32_t* __restrict__ hit, int equal_areas, int clamp_rgb,
                                                         float* __restrict__ out_rgb, float* __restrict__ out_diff, float* __restrict__ out_spec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const int p = p0 + i;
    const float PI = 3.14159265358979323846f;
    const float dconst = (float)(1.0 / 3.14159265358979323846);
    const v3 n = ld3(normal, p), al = ld3(albedo, p), ro = ld3(rough, p), fr = ld3(fresnel, p);
    const v3 surf2c = torch_normalize(-ld3(rays_d, p), 1e-6f);
    const v3 bd = V3(al.x / PI, al.y / PI, al.z / PI);
    v3 s_all = V3(0.f), s_d = V3(0.f), s_s = V3(0.f);
    for (int l = 0; l < L; l++) {
        const v3 d = ld3(dirs, l);
        const float cosine = fmaxf((d.x * n.x + d.y * n.y) + d.z * n.z, 0.0f);
        const int32_t sl = slot[(size_t)l * np + i];
        const float vis = (sl >= 0 && hit[sl]) ? 0.f : 1.f;
        const v3 sp = ggx_specular(n, surf2c, d, ro, fr);
        const v3 light = ld3(light_rgb, l) * vis;
        if (equal_areas) {
            s_d = s_d + V3(4 * PI * dconst * light.x * cosine, 4 * PI * dconst * light.y * cosine, 4 * PI * dconst * light.z * cosine);
            s_s = s_s + V3(4 * PI * sp.x * light.x * cosine, 4 * PI * sp.y * light.y * cosine, 4 * PI * sp.z * light.z * cosine);
            s_all = s_all + V3(4 * PI * (bd.x + sp.x) * light.x * cosine, 4 * PI * (bd.y + sp.y) * light.y * cosine, 4 * PI * (bd.z + sp.z) * light.z * cosine);
        } else {
            const float w = light_w[l];
            s_d = s_d + V3(dconst * light.x * cosine * w, dconst * light.y * cosine * w, dconst * light.z * cosine * w);
            s_s = s_s + V3(sp.x * light.x * cosine * w, sp.y * light.y * cosine * w, sp.z * light.z * cosine * w);
            s_all = s_all + V3((bd.x + sp.x) * light.x * cosine * w, (bd.y + sp.y) * light.y * cosine * w, (bd.z + sp.z) * light.z * cosine * w);
        }
    }
    if (equal_areas) { const float inv = (float)L; s_d = V3(s_d.x / inv, s_d.y / inv, s_d.z / inv); s_s = V3(s_s.x / inv, s_s.y / inv, s_s.z / inv); s_all = V3(s_all.x / inv, s_all.y / inv, s_all.z / inv); }
    if (clamp_rgb) s_all = V3(fminf(fmaxf(s_all.x, 0.f), 1.f), fminf(fmaxf(s_all.y, 0.f), 1.f), fminf(fmaxf(s_all.z, 0.f), 1.f));
    st3(out_rgb, p, s_all); st3(out_diff, p, s_d); st3(out_spec, p, s_s);
}

}  // namespace mr

using namespace mr;

extern "C" int mirres_dump_render(mirres_bvh_t* bvh, int n, int L, const float* pos, const float* normal, const float* albedo, const float* roughness,
                                  const float* fresnel, const float* rays_d, const float* env_map, int env_h, int env_w, const float* light_dirs,
                                  const float* light_area_weight, int equal_areas, int clamp_rgb, float* light_rgbs, float* out_rgb, float* out_diff,
                                  float* out_spec, void* stream) {
    const bool pts_ok = n == 0 || (pos && normal && albedo && roughness && fresnel && rays_d && out_rgb && out_diff && out_spec);   // an empty batch may carry null pointers
    if (!bvh || n < 0 || L <= 0 || !pts_ok || !env_map || env_h <= 0 || env_w <= 0 || !light_dirs || (!equal_areas && !light_area_weight) || !light_rgbs) {
        set_error("mirres_dump_render: bad argument"); return MIRRES_E_ARG;
    }
    if (bvh->T < 2) { set_error("mirres_dump_render: BVH not built"); return MIRRES_E_STATE; }
    hipStream_t s = (hipStream_t)stream;
    k_dump_light_rgb<<<grid_for(L, MR_BLOCK), MR_BLOCK, 0, s>>>(env_map, env_h, env_w, light_dirs, L, light_rgbs);
    MR_LAUNCH_CHECK("dump_light_rgb");
    if (n == 0) return MIRRES_OK;
    // pixel chunks of at most 2^24 (point, light) pairs: 40 bytes of pool per pair (ray, result, slot), kept in the BVH object
    size_t cap_pairs = (size_t)1 << 24;
    { const char* e = getenv("MIRRES_DUMP_CHUNK"); if (e && atoll(e) > 0) cap_pairs = (size_t)atoll(e); }   // tests force several chunks on small inputs
    int np_max = (int)(cap_pairs / (size_t)L); if (np_max < 1) np_max = 1; if (np_max > n) np_max = n;
    const size_t pairs = (size_t)np_max * (size_t)L;
    const size_t need = pairs * (sizeof(Ray) + 4 + 4) + 256;
    if (bvh->dump_pool_bytes < need) {
        if (bvh->dump_pool) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipFree(bvh->dump_pool)); bvh->dump_pool = nullptr; bvh->dump_pool_bytes = 0; }
        MR_HIP(hipMalloc(&bvh->dump_pool, need));
        bvh->dump_pool_bytes = need;
    }
    Ray* rays = reinterpret_cast<Ray*>(bvh->dump_pool);
    int32_t* hit = reinterpret_cast<int32_t*>(bvh->dump_pool + pairs * sizeof(Ray));
    int32_t* slot = hit + pairs;
    uint32_t* count = reinterpret_cast<uint32_t*>(slot + pairs);
    for (int p0 = 0; p0 < n; p0 += np_max) {
        const int np = (n - p0 < np_max) ? (n - p0) : np_max;
        const size_t tot = (size_t)np * (size_t)L;
        MR_HIP(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
        k_dump_gen<<<grid_for(tot, MR_DUMP_BLOCK), MR_DUMP_BLOCK, 0, s>>>(p0, np, L, pos, normal, light_dirs, rays, count, slot);
        MR_LAUNCH_CHECK("dump_gen");
        int rc = trace_any_front_queue(bvh, rays, count, tot, hit, s); if (rc) return rc;
        k_dump_shade<<<grid_for(np, MR_BLOCK), MR_BLOCK, 0, s>>>(p0, np, L, normal, albedo, roughness, fresnel, rays_d, light_dirs, light_area_weight, light_rgbs, slot, hit,
                                                                 equal_areas, clamp_rgb, out_rgb, out_diff, out_spec);
        MR_LAUNCH_CHECK("dump_shade");
    }
    return MIRRES_OK;
}
