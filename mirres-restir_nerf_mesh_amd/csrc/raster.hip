// raster.hip — the step in front of the path: primary visibility and attribute interpolation for the G-buffer, which the reference
// gets from nvdiffrast (dr.rasterize / dr.interpolate, nerf/renderer.py:983-998; nvdiffrast is an un-vendored dependency, readme.md). Here the
// primary rays are cast through the path's own BVH (closest hit, mirres_bvh_trace mode 2) and the hit is turned into nvdiffrast's raster record
// (u, v, depth, triangle_id + 1) so that interpolation and its gradients follow dr.interpolate's published contract:
//   out = u a[i0] + v a[i1] + (1 - u - v) a[i2],   d out / d a[i_k] = barycentric weight,   d out / d(u, v) = (a[i0] - a[i2], a[i1] - a[i2]).
// dr.texture (linear filter, clamp boundary: the jittered taps of the smoothness regularisers, :1001-1010) is k_texture below; dr.antialias
// (visibility gradients) lives in antialias.hip.
#include "engine.hpp"
#include "device_math.hpp"

namespace mr {

#define MR_BLOCK 256

// barycentrics of the hit on its triangle: the Moeller-Trumbore (u, v) of bvh_hit (weights of v1, v2) re-expressed in nvdiffrast's order (weights of v0, v1)
__global__ void __launch_bounds__(MR_BLOCK) k_rast_record(int n, const float* __restrict__ rays, const int32_t* __restrict__ hit, const float* __restrict__ t,
                                                          const int32_t* __restrict__ prim, const float* __restrict__ vert, const int32_t* __restrict__ tri,
                                                          float* __restrict__ rast) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hit[i]) {
        const int p = prim[i];
        const int32_t* ti = tri + 3 * (size_t)p;
        const v3 v0 = ld3(vert, ti[0]), E1 = ld3(vert, ti[1]) - v0, E2 = ld3(vert, ti[2]) - v0;
        const v3 o = V3(rays[8 * (size_t)i], rays[8 * (size_t)i + 1], rays[8 * (size_t)i + 2]);
        const v3 d = normalize(V3(rays[8 * (size_t)i + 4], rays[8 * (size_t)i + 5], rays[8 * (size_t)i + 6]));   // bvh_hit normalises the direction
        const v3 P = cross(d, E2);
        const float invDet = 1 / dot(E1, P);
        const v3 Tv = o - v0;
        const float u = dot(Tv, P) * invDet;
        const v3 Q = cross(Tv, E1);
        const float v = dot(d, Q) * invDet;
        r = make_float4(1.f - u - v, u, t[i], (float)(p + 1));
    }
    reinterpret_cast<float4*>(rast)[i] = r;
}

// ---- dr.rasterize(glctx, pos_clip, tri, (H, W)) (nerf/renderer.py:983): the raster record of nvdiffrast for a clip-space vertex buffer.
// Pixel (ix, iy) looks along the line NDC (x, y) = ((2 ix + 1) / W - 1, (2 iy + 1) / H - 1): its world-space pre-image, a line through the eye, is the
// primary ray, cast through the path's BVH (world space: no second hierarchy over projected
// vertices).  Record: perspective-correct barycentrics (u, v) = weights of v0, v1 — the world-space barycentrics of the hit —, z / w of the hit in clip
// space (nvdiffrast's third channel), triangle id + 1.  The ray starts where the pixel's line crosses the near plane (z_c = -w_c), so a triangle that crosses
// the near plane shows its part beyond it and hides nothing with the part in front — what clipping does for a point-sampled rasteriser; hits beyond the far
// plane give an empty record.  rast_db = (du/dX, du/dY, dv/dX, dv/dY) per pixel step, analytic: the pixel's direction is affine in NDC (x, y) and the
// barycentrics on the hit triangle's plane are ratios of linear forms of the direction (rounds 1-4: central differences half a pixel to either side).
struct Mat4 { float m[16]; };   // row-major
MR_DEV void mul4(const Mat4& M, float x, float y, float z, float w, float o[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) o[r] = ((M.m[4 * r] * x + M.m[4 * r + 1] * y) + M.m[4 * r + 2] * z) + M.m[4 * r + 3] * w;
}
// The line a pixel looks along needs only the x, y and w rows of the matrix: NDC x = x_c / w_c means (row_x - x row_w) . (P, 1) = 0, likewise for y — two planes
// through the eye (the point where x_c = y_c = w_c = 0, solved on the host in double precision); the direction is the cross product of their normals, turned
// towards increasing w.  The z row — ill-conditioned for far / near = 2 10^4, and through the inverse matrix it would spoil the directions too — enters the z / w
// output only.  `eye` travels in Mat4::m[12..14] of the second argument, rows x, y, w in m[0..11].
MR_DEV void pixel_ray(const Mat4& R, float px, float py, int W, int H, v3& o, v3& d, float* sign = nullptr) {
    const float x = (2.f * px) / W - 1.f, y = (2.f * py) / H - 1.f;
    const v3 rx = V3(R.m[0], R.m[1], R.m[2]), ry = V3(R.m[4], R.m[5], R.m[6]), rw = V3(R.m[8], R.m[9], R.m[10]);
    const v3 n1 = rx - rw * x, n2 = ry - rw * y;
    d = cross(n1, n2);
    const bool flip = dot(rw, d) < 0.f;
    if (flip) d = -d;
    if (sign) *sign = flip ? -1.f : 1.f;
    o = V3(R.m[12], R.m[13], R.m[14]);
}
// Where the line o + s d enters the clip volume through the near plane: (z_c + w_c)(P) = (row_z + row_w) . (P, 1) grows along a ray that looks into the
// scene and vanishes on the near plane.  The origin moves there when that point lies in front of the eye; a matrix whose z row does not behave like a
// projection's (z_c + w_c not growing along the ray) leaves the ray at the eye.
MR_DEV v3 near_start(const Mat4& M, v3 o, v3 d) {
    const v3 nz = V3(M.m[8] + M.m[12], M.m[9] + M.m[13], M.m[10] + M.m[14]);
    const float a = dot(nz, o) + (M.m[11] + M.m[15]), b = dot(nz, d);
    if (b > 0.f && a < 0.f) o = o + d * (-a / b);
    return o;
}
__global__ void __launch_bounds__(MR_BLOCK) k_rast_rays(Mat4 M, Mat4 R, int W, int H, float* __restrict__ rays) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    v3 o, d; pixel_ray(R, (i % W) + 0.5f, (i / W) + 0.5f, W, H, o, d);
    o = near_start(M, o, d);
    float4 a, b; a.x = o.x; a.y = o.y; a.z = o.z; a.w = 0.f; b.x = d.x; b.y = d.y; b.z = d.z; b.w = 1e7f;
    reinterpret_cast<float4*>(rays)[2 * (size_t)i] = a; reinterpret_cast<float4*>(rays)[2 * (size_t)i + 1] = b;
}
// barycentrics (weights of v0, v1) of the point where the line o + s d meets the plane of the triangle
MR_DEV void plane_bary(v3 o, v3 d, v3 v0, v3 E1, v3 E2, float& b0, float& b1) {
    const v3 P = cross(d, E2);
    const float invDet = 1 / dot(E1, P);
    const v3 Tv = o - v0;
    const float u = dot(Tv, P) * invDet;
    const float v = dot(d, cross(Tv, E1)) * invDet;
    b0 = 1.f - u - v; b1 = u;
}
__global__ void __launch_bounds__(MR_BLOCK) k_rast_record_clip(Mat4 M, Mat4 R, int W, int H, const int32_t* __restrict__ hit, const float* __restrict__ t,
                                                               const int32_t* __restrict__ prim, const float* __restrict__ vert, const int32_t* __restrict__ tri,
                                                               float* __restrict__ rast, float* __restrict__ rast_db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f), db = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hit[i] && t[i] > 0.f) {
        const int p = prim[i];
        const int32_t* ti = tri + 3 * (size_t)p;
        const v3 v0 = ld3(vert, ti[0]), E1 = ld3(vert, ti[1]) - v0, E2 = ld3(vert, ti[2]) - v0;
        const float px = (i % W) + 0.5f, py = (i / W) + 0.5f;
        v3 o, d; float sg; pixel_ray(R, px, py, W, H, o, d, &sg);
        float b0, b1; plane_bary(o, d, v0, E1, E2, b0, b1);
        const v3 P = near_start(M, o, d) + normalize(d) * t[i];  // bvh_hit normalises the direction: t is a world-space distance from where the ray started
        float c[4]; mul4(M, P.x, P.y, P.z, 1.f, c);
        const float zw = fmaxf(c[2] / c[3], -1.f);              // the ray started on the near plane: a hit there is on it, whatever the rounding says
        if (zw <= 1.f) {                                        // beyond the far plane: clipped
            r = make_float4(b0, b1, zw, (float)(p + 1));
            // d(x, y) = cross(rx - x rw, ry - y rw) = cross(rx, ry) + x cross(ry, rw) + y cross(rw, rx); with A = cross(E2, o - v0), B = cross(o - v0, E1),
            // N = cross(E2, E1): u = d.A / d.N, v = d.B / d.N, weight of v0 = 1 - u - v, of v1 = u
            const v3 rx = V3(R.m[0], R.m[1], R.m[2]), ry = V3(R.m[4], R.m[5], R.m[6]), rw = V3(R.m[8], R.m[9], R.m[10]);
            const v3 dx = cross(ry, rw) * (sg * 2.f / W), dy = cross(rw, rx) * (sg * 2.f / H);           // per pixel step, turned with d
            const v3 Tv = o - v0, A = cross(E2, Tv), B = cross(Tv, E1), N = cross(E2, E1);
            const float den = dot(d, N), inv = 1.f / den;
            const float u = dot(d, A) * inv, v = dot(d, B) * inv;
            const float ux = (dot(dx, A) - u * dot(dx, N)) * inv, uy = (dot(dy, A) - u * dot(dy, N)) * inv;
            const float vx = (dot(dx, B) - v * dot(dx, N)) * inv, vy = (dot(dy, B) - v * dot(dy, N)) * inv;
            db = make_float4(-ux - vx, -uy - vy, ux, uy);
        }
    }
    reinterpret_cast<float4*>(rast)[i] = r;
    if (rast_db) reinterpret_cast<float4*>(rast_db)[i] = db;
}

__global__ void __launch_bounds__(MR_BLOCK) k_interpolate(const float* __restrict__ attr, int C, const float* __restrict__ rast, const int32_t* __restrict__ tri, int n,
                                                          float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * C) return;
    const int i = (int)(idx / C), c = (int)(idx % C);
    const float4 r = reinterpret_cast<const float4*>(rast)[i];
    const int p = (int)r.w - 1;
    float o = 0.f;
    if (p >= 0) {
        const int32_t* ti = tri + 3 * (size_t)p;
        const float a0 = attr[(size_t)ti[0] * C + c], a1 = attr[(size_t)ti[1] * C + c], a2 = attr[(size_t)ti[2] * C + c];
        o = r.x * a0 + r.y * a1 + (1.f - r.x - r.y) * a2;
    }
    out[idx] = o;
}

__global__ void __launch_bounds__(MR_BLOCK) k_interpolate_bwd(const float* __restrict__ attr, int C, const float* __restrict__ rast, const int32_t* __restrict__ tri, int n,
                                                              const float* __restrict__ g_out, float* __restrict__ g_attr, float* __restrict__ g_uv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r = reinterpret_cast<const float4*>(rast)[i];
    const int p = (int)r.w - 1;
    float gu = 0.f, gv = 0.f;
    if (p >= 0) {
        const int32_t* ti = tri + 3 * (size_t)p;
        const float b2 = 1.f - r.x - r.y;
        for (int c = 0; c < C; c++) {
            const float g = g_out[(size_t)i * C + c];
            if (g_attr) {
                atomicAdd(&g_attr[(size_t)ti[0] * C + c], r.x * g); atomicAdd(&g_attr[(size_t)ti[1] * C + c], r.y * g); atomicAdd(&g_attr[(size_t)ti[2] * C + c], b2 * g);
            }
            if (g_uv) {
                const float a0 = attr[(size_t)ti[0] * C + c], a1 = attr[(size_t)ti[1] * C + c], a2 = attr[(size_t)ti[2] * C + c];
                gu += g * (a0 - a2); gv += g * (a1 - a2);
            }
        }
    }
    if (g_uv) { g_uv[2 * (size_t)i] = gu; g_uv[2 * (size_t)i + 1] = gv; }
}

// dr.texture(tex[H,W,C], uv[n,2], filter_mode='linear', boundary_mode='clamp') (nerf/renderer.py:1004,1008): texel centres at (i + 0.5) / W, the
// four taps clamped to the edge. Same arithmetic as F.grid_sample(bilinear, padding 'border', align_corners=False) on uv * 2 - 1.
MR_DEV void tex_taps(float u, float v, int W, int H, int& x0, int& x1, int& y0, int& y1, float& fx, float& fy) {
    float x = u * W - 0.5f, y = v * H - 0.5f;
    x = fminf(fmaxf(x, 0.f), (float)(W - 1)); y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    const float xf = floorf(x), yf = floorf(y);
    x0 = (int)xf; y0 = (int)yf; x1 = min(x0 + 1, W - 1); y1 = min(y0 + 1, H - 1);
    fx = x - xf; fy = y - yf;
}
__global__ void __launch_bounds__(MR_BLOCK) k_texture(const float* __restrict__ tex, int H, int W, int C, const float* __restrict__ uv, int n, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * C) return;
    const int i = (int)(idx / C), c = (int)(idx % C);
    int x0, x1, y0, y1; float fx, fy;
    tex_taps(uv[2 * (size_t)i], uv[2 * (size_t)i + 1], W, H, x0, x1, y0, y1, fx, fy);
    const float t00 = tex[((size_t)y0 * W + x0) * C + c], t01 = tex[((size_t)y0 * W + x1) * C + c];
    const float t10 = tex[((size_t)y1 * W + x0) * C + c], t11 = tex[((size_t)y1 * W + x1) * C + c];
    out[idx] = (t00 * (1.f - fx) + t01 * fx) * (1.f - fy) + (t10 * (1.f - fx) + t11 * fx) * fy;
}
__global__ void __launch_bounds__(MR_BLOCK) k_texture_bwd(int H, int W, int C, const float* __restrict__ uv, int n, const float* __restrict__ g_out, float* __restrict__ g_tex) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * C) return;
    const int i = (int)(idx / C), c = (int)(idx % C);
    int x0, x1, y0, y1; float fx, fy;
    tex_taps(uv[2 * (size_t)i], uv[2 * (size_t)i + 1], W, H, x0, x1, y0, y1, fx, fy);
    const float g = g_out[idx];
    atomicAdd(&g_tex[((size_t)y0 * W + x0) * C + c], g * (1.f - fx) * (1.f - fy)); atomicAdd(&g_tex[((size_t)y0 * W + x1) * C + c], g * fx * (1.f - fy));
    atomicAdd(&g_tex[((size_t)y1 * W + x0) * C + c], g * (1.f - fx) * fy); atomicAdd(&g_tex[((size_t)y1 * W + x1) * C + c], g * fx * fy);
}

}  // namespace mr

using namespace mr;

extern "C" int mirres_texture2d(const float* tex, int H, int W, int C, const float* uv, int n, float* out, void* stream) {
    if (H <= 0 || W <= 0 || C <= 0 || n < 0 || !tex || (n > 0 && (!uv || !out))) { set_error("mirres_texture2d: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    k_texture<<<grid_for((size_t)n * C, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(tex, H, W, C, uv, n, out);
    MR_LAUNCH_CHECK("texture2d");
    return MIRRES_OK;
}
extern "C" int mirres_texture2d_bwd(int H, int W, int C, const float* uv, int n, const float* g_out, float* g_tex, void* stream) {
    if (H <= 0 || W <= 0 || C <= 0 || n < 0 || !g_tex || (n > 0 && (!uv || !g_out))) { set_error("mirres_texture2d_bwd: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    k_texture_bwd<<<grid_for((size_t)n * C, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(H, W, C, uv, n, g_out, g_tex);
    MR_LAUNCH_CHECK("texture2d_bwd");
    return MIRRES_OK;
}

extern "C" int mirres_raster_raycast(mirres_bvh_t* bvh, const float* rays, int n, const float* vert, const int32_t* tri, float* rast, void* stream) {
    if (!bvh || n < 0 || (n > 0 && (!rays || !vert || !tri || !rast))) { set_error("mirres_raster_raycast: bad argument"); return MIRRES_E_ARG; }
    if (bvh->T < 2) { set_error("mirres_raster_raycast: BVH not built"); return MIRRES_E_STATE; }
    if (n == 0) return MIRRES_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t need = (size_t)n * 12 + 256;          // hit, t, prim of the batch (kept in the BVH object's scratch pool)
    if (bvh->dump_pool_bytes < need) {
        if (bvh->dump_pool) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipFree(bvh->dump_pool)); bvh->dump_pool = nullptr; bvh->dump_pool_bytes = 0; }
        MR_HIP(hipMalloc(&bvh->dump_pool, need));
        bvh->dump_pool_bytes = need;
    }
    int32_t* hit = reinterpret_cast<int32_t*>(bvh->dump_pool); float* t = reinterpret_cast<float*>(hit + n); int32_t* prim = reinterpret_cast<int32_t*>(t + n);
    int rc = mirres_bvh_trace(bvh, rays, n, 2, hit, t, nullptr, nullptr, prim, nullptr, stream); if (rc) return rc;
    k_rast_record<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, s>>>(n, rays, hit, t, prim, vert, tri, rast);
    MR_LAUNCH_CHECK("raster_raycast");
    return MIRRES_OK;
}

extern "C" int mirres_rasterize(mirres_bvh_t* bvh, const float* vert, const int32_t* tri, const float* h_mvp, const float* h_eye, int W, int H,
                                float* rast, float* rast_db, void* stream) {
    if (!bvh || !vert || !tri || !h_mvp || !h_eye || W <= 0 || H <= 0 || !rast) { set_error("mirres_rasterize: bad argument"); return MIRRES_E_ARG; }
    if (bvh->T < 2) { set_error("mirres_rasterize: BVH not built"); return MIRRES_E_STATE; }
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)W * H;
    const size_t need = n * (32 + 12) + 256;           // rays + hit, t, prim of the frame (kept in the BVH object's scratch pool)
    if (bvh->dump_pool_bytes < need) {
        if (bvh->dump_pool) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipFree(bvh->dump_pool)); bvh->dump_pool = nullptr; bvh->dump_pool_bytes = 0; }
        MR_HIP(hipMalloc(&bvh->dump_pool, need));
        bvh->dump_pool_bytes = need;
    }
    float* rays = reinterpret_cast<float*>(bvh->dump_pool);
    int32_t* hit = reinterpret_cast<int32_t*>(rays + 8 * n); float* t = reinterpret_cast<float*>(hit + n); int32_t* prim = reinterpret_cast<int32_t*>(t + n);
    Mat4 M, R;
    for (int k = 0; k < 16; k++) M.m[k] = h_mvp[k];
    for (int k = 0; k < 4; k++) { R.m[k] = h_mvp[k]; R.m[4 + k] = h_mvp[4 + k]; R.m[8 + k] = h_mvp[12 + k]; }     // rows x, y, w
    R.m[12] = h_eye[0]; R.m[13] = h_eye[1]; R.m[14] = h_eye[2]; R.m[15] = 0.f;
    k_rast_rays<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, s>>>(M, R, W, H, rays);
    int rc = mirres_bvh_trace(bvh, rays, (int)n, 4, hit, t, nullptr, nullptr, prim, nullptr, stream); if (rc) return rc;   // mode 4: the ray starts on the near plane, maybe inside the mesh
    k_rast_record_clip<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, s>>>(M, R, W, H, hit, t, prim, vert, tri, rast, rast_db);
    MR_LAUNCH_CHECK("rasterize");
    return MIRRES_OK;
}

extern "C" int mirres_interpolate(const float* attr, int C, const float* rast, const int32_t* tri, int n, float* out, void* stream) {
    if (C <= 0 || n < 0 || (n > 0 && (!attr || !rast || !tri || !out))) { set_error("mirres_interpolate: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    k_interpolate<<<grid_for((size_t)n * C, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(attr, C, rast, tri, n, out);
    MR_LAUNCH_CHECK("interpolate");
    return MIRRES_OK;
}

extern "C" int mirres_interpolate_bwd(const float* attr, int C, const float* rast, const int32_t* tri, int n, const float* g_out, float* g_attr, float* g_uv, void* stream) {
    if (C <= 0 || n < 0 || (n > 0 && (!attr || !rast || !tri || !g_out)) || (!g_attr && !g_uv)) { set_error("mirres_interpolate_bwd: bad argument"); return MIRRES_E_ARG; }
    if (n == 0) return MIRRES_OK;
    k_interpolate_bwd<<<grid_for(n, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(attr, C, rast, tri, n, g_out, g_attr, g_uv);
    MR_LAUNCH_CHECK("interpolate_bwd");
    return MIRRES_OK;
}
