// antialias.hip — dr.antialias (nerf/renderer.py:1184-1206): analytic edge antialiasing of the stage-1 outputs, the operator that gives the
// image loss a gradient w.r.t. vertex POSITIONS through visibility (the reference trains `vertices_offsets` with it).
//
// nvdiffrast is an un-vendored, un-pinned dependency of the reference (readme.md:33), so this restates its published algorithm (Laine et al.,
// "Modular Primitives for High-Performance Differentiable Rendering", 2020, section 3.4) — PARITY UNPINNED:
//   * every pair of horizontally / vertically adjacent pixels whose raster records name different triangles is a potential discontinuity;
//   * of the two, the triangle nearer to the camera is examined (a background pixel defers to its neighbour); call its pixel p0, the other p1;
//   * an edge of that triangle is a SILHOUETTE edge when no other triangle shares it, or when the triangle across it lies on the same side of
//     the edge in screen space (a fold);
//   * if a silhouette edge crosses the segment between the two pixel centres at fraction u in [0, 1] (from p0), the triangle covers the part
//     [0, u] of the segment: alpha = u - 1/2;  alpha > 0: p1 is partly covered,  out[p1] += alpha (in[p0] - in[p1]);
//                                              alpha < 0: p0 is partly uncovered, out[p0] += -alpha (in[p1] - in[p0]);
//   * the blend weights are differentiable in the clip-space positions of the edge's two vertices (perspective division included) — that is the
//     visibility gradient; `pos_gradient_boost` scales it.
// MI355X form: ONE thread per pixel gathers the (at most four) contributions it receives, in a fixed order — no atomics and a deterministic
// image (nvdiffrast scatters with atomicAdd); the backward pass gathers the colour gradient the same way and scatters only the position
// gradient (one pair = one thread, the pixel with the lower index). Both evaluations of a pair see the same ordered pair, hence the same alpha.
// Topology (which vertex lies across each edge) is an input: i32[T,3], entry k of triangle t = the vertex opposite edge (v_k, v_{k+1}) in the
// neighbouring triangle, -1 on a boundary; it depends on the index buffer only and is built once per mesh by the host side.
#include "engine.hpp"
#include "device_math.hpp"

namespace mr {

#define MR_BLOCK 256

struct AAPair { int p0, p1, va, vb; float alpha; float ad, ao, bd, bo; int axis; float s; bool ok; };

MR_DEV bool aa_project(const float* __restrict__ pos, int v, float half_w, float half_h, float cx, float cy, float& x, float& y) {
    const float4 c = reinterpret_cast<const float4*>(pos)[v];
    if (!(c.w > 0.f)) return false;                                   // behind the camera plane: no screen-space statement about this edge
    const float iw = 1.0f / c.w;
    x = (c.x * iw + 1.0f) * half_w - cx; y = (c.y * iw + 1.0f) * half_h - cy;   // pixel units relative to the centre of p0
    return true;
}

// (lo, hi): the pair in index order; axis 0: hi = lo + 1 (same row), axis 1: hi = lo + W
MR_DEV AAPair aa_analyse(int W, int H, const float* __restrict__ rast, const float* __restrict__ pos, const int32_t* __restrict__ tri, const int32_t* __restrict__ opp,
                         int lo, int hi, int axis) {
    AAPair r; r.ok = false; r.alpha = 0.f; r.axis = axis;
    const float4 r0 = reinterpret_cast<const float4*>(rast)[lo], r1 = reinterpret_cast<const float4*>(rast)[hi];
    const int t0 = (int)r0.w - 1, t1 = (int)r1.w - 1;
    if (t0 == t1) return r;
    bool first = t0 >= 0;                                             // the examined triangle: the one nearer to the camera, background defers
    if (t0 >= 0 && t1 >= 0) first = !(r1.z < r0.z);
    const int t = first ? t0 : t1;
    r.p0 = first ? lo : hi; r.p1 = first ? hi : lo;
    r.s = first ? 1.0f : -1.0f;                                       // direction from p0 to p1 along the axis
    const float cx = (float)(r.p0 % W) + 0.5f, cy = (float)(r.p0 / W) + 0.5f;
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    const int32_t* ti = tri + 3 * (size_t)t;
    const int vi[3] = {ti[0], ti[1], ti[2]};
    float px[3], py[3];
#pragma unroll
    for (int k = 0; k < 3; k++) if (!aa_project(pos, vi[k], hw, hh, cx, cy, px[k], py[k])) return r;
    float best_u = 2.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int ka = k, kb = (k + 1) % 3, kc = (k + 2) % 3;
        const float ax = px[ka], ay = py[ka], bx = px[kb], by = py[kb];
        // silhouette?
        const int o = opp[3 * (size_t)t + k];
        if (o >= 0) {
            float ox, oy;
            if (!aa_project(pos, o, hw, hh, cx, cy, ox, oy)) continue;
            const float ex = bx - ax, ey = by - ay;
            const float sc = ex * (py[kc] - ay) - ey * (px[kc] - ax), so = ex * (oy - ay) - ey * (ox - ax);
            if (!((sc > 0.f) == (so > 0.f))) continue;               // the neighbour continues the surface on the other side: not a silhouette
        }
        const float ad = axis ? ay : ax, ao = axis ? ax : ay, bd = axis ? by : bx, bo = axis ? bx : by;
        if ((ao > 0.f) == (bo > 0.f)) continue;                       // the edge does not cross the line through the two pixel centres
        const float xc = (ad * bo - bd * ao) / (bo - ao);
        const float u = r.s * xc;
        if (!(u >= 0.f && u <= 1.0f)) continue;
        if (u < best_u) { best_u = u; r.va = vi[ka]; r.vb = vi[kb]; r.ad = ad; r.ao = ao; r.bd = bd; r.bo = bo; }
    }
    if (best_u > 1.0f) return r;
    r.alpha = best_u - 0.5f; r.ok = true;
    return r;
}

// weight with which pixel `me` receives (in[other] - in[me]) from the pair
MR_DEV float aa_weight(const AAPair& a, int me) {
    if (!a.ok) return 0.f;
    if (me == a.p1) return a.alpha > 0.f ? a.alpha : 0.f;
    return a.alpha < 0.f ? -a.alpha : 0.f;
}

__global__ void __launch_bounds__(MR_BLOCK) k_aa_fwd(int W, int H, int C, const float* __restrict__ color, const float* __restrict__ rast, const float* __restrict__ pos,
                                                     const int32_t* __restrict__ tri, const int32_t* __restrict__ opp, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W * H) return;
    const int x = p % W, y = p / W;
    int nb[4]; float w[4];
    // fixed order: left, right, up, down
    nb[0] = x > 0 ? p - 1 : -1; nb[1] = x + 1 < W ? p + 1 : -1; nb[2] = y > 0 ? p - W : -1; nb[3] = y + 1 < H ? p + W : -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w[k] = 0.f;
        if (nb[k] >= 0) { const int lo = min(p, nb[k]), hi = max(p, nb[k]); w[k] = aa_weight(aa_analyse(W, H, rast, pos, tri, opp, lo, hi, k >> 1), p); }
    }
    for (int c = 0; c < C; c++) {
        const float me = color[(size_t)p * C + c];
        float o = me;
#pragma unroll
        for (int k = 0; k < 4; k++) if (w[k] != 0.f) o += w[k] * (color[(size_t)nb[k] * C + c] - me);
        out[(size_t)p * C + c] = o;
    }
}

// colour gradient (gather) and position gradient (scatter, one thread per pair = the lower-index pixel of the right / down pairs)
__global__ void __launch_bounds__(MR_BLOCK) k_aa_bwd(int W, int H, int C, const float* __restrict__ color, const float* __restrict__ rast, const float* __restrict__ pos,
                                                     const int32_t* __restrict__ tri, const int32_t* __restrict__ opp, const float* __restrict__ g_out,
                                                     float* __restrict__ g_color, float* __restrict__ g_pos, float boost) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W * H) return;
    const int x = p % W, y = p / W;
    int nb[4]; float w_me[4], w_nb[4];
    nb[0] = x > 0 ? p - 1 : -1; nb[1] = x + 1 < W ? p + 1 : -1; nb[2] = y > 0 ? p - W : -1; nb[3] = y + 1 < H ? p + W : -1;
    AAPair own[2];   // the right (k = 1) and down (k = 3) pairs: this thread owns their position gradient
    own[0].ok = false; own[1].ok = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w_me[k] = 0.f; w_nb[k] = 0.f;
        if (nb[k] >= 0) {
            const int lo = min(p, nb[k]), hi = max(p, nb[k]);
            const AAPair a = aa_analyse(W, H, rast, pos, tri, opp, lo, hi, k >> 1);
            w_me[k] = aa_weight(a, p); w_nb[k] = aa_weight(a, nb[k]);
            if (k & 1) own[k >> 1] = a;
        }
    }
    if (g_color) {
        for (int c = 0; c < C; c++) {
            // out[p] = in[p] (1 - sum w_me) + sum w_me in[nb];   out[nb] = ... + w_nb in[p]
            float g = g_out[(size_t)p * C + c];
            float acc = g;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (w_me[k] != 0.f) acc -= w_me[k] * g;
                if (w_nb[k] != 0.f) acc += w_nb[k] * g_out[(size_t)nb[k] * C + c];
            }
            g_color[(size_t)p * C + c] = acc;
        }
    }
    if (g_pos) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const AAPair& a = own[j];
            if (!a.ok) continue;
            // d L / d alpha
            float ga = 0.f;
            for (int c = 0; c < C; c++) {
                const float c0 = color[(size_t)a.p0 * C + c], c1 = color[(size_t)a.p1 * C + c];
                if (a.alpha > 0.f) ga += g_out[(size_t)a.p1 * C + c] * (c0 - c1);
                else if (a.alpha < 0.f) ga -= g_out[(size_t)a.p0 * C + c] * (c1 - c0);
            }
            if (ga == 0.f) continue;
            // alpha = s * xc - 1/2,  xc = (ad bo - bd ao) / (bo - ao)
            const float D = a.bo - a.ao, iD = 1.0f / D, iD2 = iD * iD;
            const float g_xc = ga * a.s * boost;
            const float g_ad = g_xc * a.bo * iD, g_bd = -g_xc * a.ao * iD;
            const float g_ao = g_xc * a.bo * (a.ad - a.bd) * iD2, g_bo = g_xc * a.ao * (a.bd - a.ad) * iD2;
            // pixel coordinate -> clip:  X = (x / w + 1) W/2  =>  dX/dx = W / (2 w), dX/dw = -x W / (2 w^2)
            const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
            const int vs[2] = {a.va, a.vb};
            const float gd[2] = {g_ad, g_bd}, go[2] = {g_ao, g_bo};
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float4 cpos = reinterpret_cast<const float4*>(pos)[vs[e]];
                const float iw = 1.0f / cpos.w;
                const float gX = a.axis ? go[e] : gd[e], gY = a.axis ? gd[e] : go[e];   // gradient w.r.t. the pixel-space x and y of the vertex
                const float gx = gX * hw * iw, gy = gY * hh * iw;
                const float gw = -(gX * hw * cpos.x + gY * hh * cpos.y) * iw * iw;
                atomicAdd(&g_pos[4 * (size_t)vs[e]], gx); atomicAdd(&g_pos[4 * (size_t)vs[e] + 1], gy); atomicAdd(&g_pos[4 * (size_t)vs[e] + 3], gw);
            }
        }
    }
}

}  // namespace mr

using namespace mr;

extern "C" int mirres_antialias(int W, int H, int C, const float* color, const float* rast, const float* pos_clip, const int32_t* tri, const int32_t* opp,
                                float* out, void* stream) {
    if (W <= 0 || H <= 0 || C <= 0 || !color || !rast || !pos_clip || !tri || !opp || !out) { set_error("mirres_antialias: bad argument"); return MIRRES_E_ARG; }
    k_aa_fwd<<<grid_for((size_t)W * H, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(W, H, C, color, rast, pos_clip, tri, opp, out);
    MR_LAUNCH_CHECK("antialias");
    return MIRRES_OK;
}

extern "C" int mirres_antialias_bwd(int W, int H, int C, const float* color, const float* rast, const float* pos_clip, const int32_t* tri, const int32_t* opp,
                                    const float* g_out, float* g_color, float* g_pos, float pos_gradient_boost, void* stream) {
    if (W <= 0 || H <= 0 || C <= 0 || !color || !rast || !pos_clip || !tri || !opp || !g_out || (!g_color && !g_pos)) {
        set_error("mirres_antialias_bwd: bad argument"); return MIRRES_E_ARG;
    }
    k_aa_bwd<<<grid_for((size_t)W * H, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(W, H, C, color, rast, pos_clip, tri, opp, g_out, g_color, g_pos, pos_gradient_boost);
    MR_LAUNCH_CHECK("antialias_bwd");
    return MIRRES_OK;
}
