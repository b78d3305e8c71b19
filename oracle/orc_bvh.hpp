// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header). PARITY UNPINNED.
// LBVH build + traversal restated from nerf/bvhworkers/*.slang and utils/helperDi.slang.
#pragma once
#include "orc_math.hpp"
#include <vector>
#include <algorithm>

namespace orc {

// ---- nerf/bvhworkers/lbvh_morton_codes.slang:24-42
static inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton3d(float x, float y, float z) {
    x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    y = fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
    z = fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    uint32_t xx = expand_bits((uint32_t)x), yy = expand_bits((uint32_t)y), zz = expand_bits((uint32_t)z);
    return xx * 4 + yy * 2 + zz;
}

// ---- lbvh_hierarchy.slang:31-53
static inline int find_msb(uint32_t v) {
    if (v == 0) return -1;
    int msb = 31;
    while (!((v >> msb) & 1u)) msb--;
    return msb;
}
static inline int delta(int i, uint32_t codeI, int j, uint32_t n, const int32_t* sorted) {
    if (j < 0 || (uint32_t)j > n - 1) return -1;
    uint32_t codeJ = (uint32_t)sorted[2 * j];
    if (codeI == codeJ) return 32 + 31 - find_msb((uint32_t)i ^ (uint32_t)j);  // sorted positions, :47-48
    return 31 - find_msb(codeI ^ codeJ);
}

struct BuildStats { int max_height; };

// Restates restirbvhWorker.update_bvh (nerf/renderer_restir.py:25-89) and its 7 kernels.
// info  int32[2T-1,3] (left,right,prim)   aabb  float[2T-1,6]   sorted int32[T,2] (code, elementIdx)
// The seven build kernels, one function each, so that the reference's own driver (restirbvhWorker.update_bvh, renderer_restir.py:25-89, executed
// from its source by tests/golden/gen_reference_loop.py) can launch them in its order; bvh_build below is the oracle's restatement of that driver.
// `cons` = g_lbvh_construction_infos i32[2T-1,2]: (parent, unused).

// generateElements  get_elements.slang:3-39
static inline void bvh_elements(const float* vert, const int32_t* tri, int T, int32_t* ele_prim, float* ele) {
    for (int p = 0; p < T; p++) {
        float mn[3] = {1e9f, 1e9f, 1e9f}, mx[3] = {-1e9f, -1e9f, -1e9f};
        for (int i = 0; i < 3; i++) {
            int vi = tri[3 * p + i];
            for (int k = 0; k < 3; k++) {
                float v = vert[3 * vi + k];
                mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v);
            }
        }
        for (int k = 0; k < 3; k++) {
            ele[6 * p + k] = fminf(mn[k], mx[k]);
            ele[6 * p + 3 + k] = fmaxf(mn[k], mx[k]);
        }
        if (ele_prim) ele_prim[p] = p;
    }
}
// morton_codes  lbvh_morton_codes.slang:46-80
static inline void bvh_morton(int T, const float* gmin, const float* gmax, const float* ele, int32_t* codes) {
    for (int g = 0; g < T; g++) {
        float c[3];
        for (int k = 0; k < 3; k++) {
            float mn = ele[6 * g + k], mx = ele[6 * g + 3 + k];
            float center = mn + 0.5f * (mx - mn);
            c[k] = (center - gmin[k]) / (gmax[k] - gmin[k]);
        }
        codes[2 * g] = (int32_t)morton3d(c[0], c[1], c[2]);
        codes[2 * g + 1] = g;
    }
}
// radix_sort  lbvh_single_radixsort.slang:28-138 : stable LSD, 4 x 8 bit, ping-pong, result back in `a`
static inline void bvh_radix_sort(int T, int32_t* a, int32_t* b) {
    for (int it = 0; it < 4; it++) {
        int shift = 8 * it;
        const int32_t* src = (it % 2 == 0) ? a : b;
        int32_t* dst = (it % 2 == 0) ? b : a;
        uint32_t hist[256] = {0};
        for (int i = 0; i < T; i++) hist[((uint32_t)src[2 * i] >> shift) & 255u]++;
        uint32_t off[256]; uint32_t s = 0;
        for (int k = 0; k < 256; k++) { off[k] = s; s += hist[k]; }
        for (int i = 0; i < T; i++) {
            uint32_t bin = ((uint32_t)src[2 * i] >> shift) & 255u;
            uint32_t o = off[bin]++;
            dst[2 * o] = src[2 * i]; dst[2 * o + 1] = src[2 * i + 1];
        }
    }
}
// hierarchy  lbvh_hierarchy.slang:111-245
static inline void bvh_hierarchy(int T, const int32_t* ele_prim, const float* ele, const int32_t* sorted, int32_t* info, float* aabb, int32_t* cons) {
    const int LEAF = T - 1;
    for (int g = 0; g < T; g++) {
        int e = sorted[2 * g + 1];
        info[3 * (LEAF + g) + 0] = 0; info[3 * (LEAF + g) + 1] = 0; info[3 * (LEAF + g) + 2] = ele_prim ? ele_prim[e] : e;
        for (int k = 0; k < 6; k++) aabb[6 * (LEAF + g) + k] = ele[6 * e + k];
    }
    for (int g = 0; g < T - 1; g++) {
        // determineRange :55-83
        uint32_t code = (uint32_t)sorted[2 * g];
        int dL = delta(g, code, g - 1, (uint32_t)T, sorted), dR = delta(g, code, g + 1, (uint32_t)T, sorted);
        int d = (dR >= dL) ? 1 : -1;
        int dMin = std::min(dL, dR);
        int lMax = 2;
        while (delta(g, code, g + lMax * d, (uint32_t)T, sorted) > dMin) lMax <<= 1;
        int l = 0;
        for (int t = lMax >> 1; t > 0; t >>= 1)
            if (delta(g, code, g + (l + t) * d, (uint32_t)T, sorted) > dMin) l += t;
        int j = g + l * d;
        int first = std::min(g, j), last = std::max(g, j);
        // findSplit :85-109
        uint32_t firstCode = (uint32_t)sorted[2 * first];
        int common = delta(first, firstCode, last, (uint32_t)T, sorted);
        int split = first, stride = last - first;
        do {
            stride = (stride + 1) >> 1;
            int ns = split + stride;
            if (ns < last) {
                int sp = delta(first, firstCode, ns, (uint32_t)T, sorted);
                if (sp > common) split = ns;
            }
        } while (stride > 1);
        int cA = (split == first) ? LEAF + split : split;
        int cB = (split + 1 == last) ? LEAF + split + 1 : split + 1;
        info[3 * g + 0] = cA; info[3 * g + 1] = cB; info[3 * g + 2] = 0;
        for (int k = 0; k < 3; k++) { aabb[6 * g + k] = 1e9f; aabb[6 * g + 3 + k] = -1e9f; }
        cons[2 * cA] = g; cons[2 * cB] = g;
    }
    cons[0] = 0;
}
// get_bvh_height  lbvh_bounding_boxes.slang:151-173
static inline void bvh_heights(int T, const int32_t* cons, int32_t* heights) {
    const int LEAF = T - 1;
    for (int g = 0; g < T; g++) {
        uint32_t n = (uint32_t)cons[2 * (LEAF + g)]; int h = 0;
        while (n != 0) { h++; n = (uint32_t)cons[2 * n]; }
        heights[g] = h;
    }
}
// get_bbox, one pass  :175-298
static inline void bvh_bbox_pass(int T, int eh, const int32_t* info, float* aabb, const int32_t* cons) {
    const int LEAF = T - 1;
    for (int g = 0; g < T; g++) {
        uint32_t n = (uint32_t)cons[2 * (LEAF + g)]; int h = 0;
        while (true) {
            if (n == 0) break;
            h++;
            if (h > eh) break;
            if (h == eh) {
                int L = info[3 * n], R = info[3 * n + 1];
                float mnA[3], mxA[3], mnB[3], mxB[3];
                for (int k = 0; k < 3; k++) {
                    mnA[k] = aabb[6 * L + k]; mxA[k] = aabb[6 * L + 3 + k];
                    mnB[k] = aabb[6 * R + k]; mxB[k] = aabb[6 * R + 3 + k];
                    if (L == 0) { mnA[k] = 1e9f; mxA[k] = -1e9f; }
                    if (R == 0) { mnB[k] = 1e9f; mxB[k] = -1e9f; }
                    aabb[6 * n + k] = fminf(mnA[k], mnB[k]);
                    aabb[6 * n + 3 + k] = fmaxf(mxA[k], mxB[k]);
                }
                break;
            }
            n = (uint32_t)cons[2 * n];
        }
    }
}
// set_root  :300-389
static inline void bvh_set_root(const int32_t* info, float* aabb) {
    int L = info[0], R = info[1];
    for (int k = 0; k < 3; k++) {
        aabb[k] = fminf(aabb[6 * L + k], aabb[6 * R + k]);
        aabb[3 + k] = fmaxf(aabb[6 * L + 3 + k], aabb[6 * R + 3 + k]);
    }
}

// restirbvhWorker.update_bvh (renderer_restir.py:25-89): the driver of the seven kernels
static inline void bvh_build(const float* vert, int V, const int32_t* tri, int T,
                             int32_t* info, float* aabb, int32_t* sorted_out, BuildStats* st) {
    (void)V;
    const int N = 2 * T - 1;
    std::vector<float> ele(6 * (size_t)T);
    std::vector<int32_t> ele_prim((size_t)T);
    bvh_elements(vert, tri, T, ele_prim.data(), ele.data());
    // scene extent  renderer_restir.py:34-40
    float gmin[3] = {INFINITY, INFINITY, INFINITY}, gmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int p = 0; p < T; p++)
        for (int k = 0; k < 3; k++) {
            gmin[k] = fminf(gmin[k], ele[6 * p + k]);
            gmax[k] = fmaxf(gmax[k], ele[6 * p + 3 + k]);
        }
    std::vector<int32_t> a(2 * (size_t)T), b(2 * (size_t)T, 0);
    bvh_morton(T, gmin, gmax, ele.data(), a.data());
    bvh_radix_sort(T, a.data(), b.data());
    if (sorted_out) std::memcpy(sorted_out, a.data(), sizeof(int32_t) * 2 * (size_t)T);
    std::vector<int32_t> cons(2 * (size_t)N, 0);
    bvh_hierarchy(T, ele_prim.data(), ele.data(), a.data(), info, aabb, cons.data());
    std::vector<int32_t> heights((size_t)T);
    bvh_heights(T, cons.data(), heights.data());
    int hmax = 0;
    for (int g = 0; g < T; g++) hmax = std::max(hmax, heights[g]);
    if (st) st->max_height = hmax;
    for (int eh = 1; eh <= hmax; eh++) bvh_bbox_pass(T, eh, info, aabb, cons.data());   // renderer_restir.py:78-83
    bvh_set_root(info, aabb);
}

// ---------------------------------------------------------------- traversal  (utils/helperDi.slang:136-395)
struct Bvh {
    const int32_t* info; const float* aabb; const float* vert; const int32_t* tri;
};
struct TraceCounters { uint32_t popped, entered, leaves, overflow, max_count; };   // max_count: most entries the 64-entry stack ever held (helperDi.slang:136)

// helperDi.slang:149-170
static inline bool aabb_hit(f3 o, f3 d, float t_min, float t_max, const float* bb) {
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (int i = 0; i < 3; ++i) {
        float di = dd[i];
        if (di == 0.f) di = 0.000001f;
        float inv = 1.0f / di;
        float t0 = (bb[i] - oo[i]) * inv;
        float t1 = (bb[3 + i] - oo[i]) * inv;
        if (inv < 0.0f) { float tmp = t1; t1 = t0; t0 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

// helperDi.slang:172-195 / :277-310 (with_normal)
static inline bool triangle_hit(f3 o, f3 d, f3 v0, f3 v1, f3 v2, float& t_hit, bool want_normal, f3& normal) {
    const float eps = 1e-15f;
    f3 E1 = v1 - v0, E2 = v2 - v0;
    f3 P = cross(d, E2);
    float det = dot(E1, P);
    if (det > -eps && det < eps) return false;
    float invDet = 1 / det;
    f3 Tv = o - v0;
    float u = dot(Tv, P) * invDet;
    if (u < 0 || u > 1) return false;
    f3 Q = cross(Tv, E1);
    float v = dot(d, Q) * invDet;
    if (v < 0 || u + v > 1) return false;
    t_hit = dot(E2, Q) * invDet;
    if (want_normal) {
        f3 fn = normalize(cross(E1, E2));
        float r = 1.0f - u - v;
        f3 n = u * fn + v * fn + r * fn;
        if (dot(-d, n) < 0) n = -n;
        normal = normalize(n);
    }
    return true;
}

struct HitResult { bool hit; float t; f3 pos; f3 normal; int prim; };

// bvh_hit (:197-274) and bvh_hit_with_normal (:313-395). `prim` is the build-defined hit index
// (SURVEY §8 a-9): primitiveIdx of the last leaf whose test satisfied t <= closest; -1 on miss.
static inline HitResult bvh_hit(const Bvh& B, f3 o, f3 d, float t_min, float t_max, bool want_normal, TraceCounters* tc) {
    d = normalize(d);
    struct Node { int index, left, right; uint32_t prim; };
    Node stack[64];
    int count = 0;
    Node root = {0, B.info[0], B.info[1], (uint32_t)B.info[2]};
    stack[count++] = root;
    float closest = t_max;
    HitResult r; r.hit = false; r.t = 0.f; r.pos = mk3(0.f); r.normal = mk3(1.f); r.prim = -1;
    while (count > 0) {
        Node n = stack[--count];
        if (tc) tc->popped++;
        if (!aabb_hit(o, d, t_min, closest, B.aabb + 6 * (size_t)n.index)) continue;
        if (n.left != 0 && n.right != 0) {
            if (tc) tc->entered++;
            if (count + 2 > 64) { if (tc) tc->overflow++; continue; }  // reference has no check (UB); flagged
            Node l = {n.left, B.info[3 * n.left], B.info[3 * n.left + 1], (uint32_t)B.info[3 * n.left + 2]};
            Node rr = {n.right, B.info[3 * n.right], B.info[3 * n.right + 1], (uint32_t)B.info[3 * n.right + 2]};
            stack[count++] = l; stack[count++] = rr;
            if (tc && (uint32_t)count > tc->max_count) tc->max_count = (uint32_t)count;
        } else if (n.left == 0 && n.right == 0) {
            if (tc) tc->leaves++;
            const int32_t* ti = B.tri + 3 * (size_t)n.prim;
            f3 v0 = mk3(B.vert[3 * ti[0]], B.vert[3 * ti[0] + 1], B.vert[3 * ti[0] + 2]);
            f3 v1 = mk3(B.vert[3 * ti[1]], B.vert[3 * ti[1] + 1], B.vert[3 * ti[1] + 2]);
            f3 v2 = mk3(B.vert[3 * ti[2]], B.vert[3 * ti[2] + 1], B.vert[3 * ti[2] + 2]);
            float now_t = 0.f; f3 now_n = mk3(1.f);
            bool hit = triangle_hit(o, d, v0, v1, v2, now_t, want_normal, now_n);
            closest = hit ? fminf(now_t, closest) : closest;
            if (hit) {
                r.hit = true; r.t = closest; r.pos = o + closest * d;
                if (now_t <= closest) { r.normal = now_n; r.prim = (int)n.prim; }
            }
        }
    }
    return r;
}

}  // namespace orc
