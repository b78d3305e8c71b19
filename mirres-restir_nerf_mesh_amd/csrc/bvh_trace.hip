// bvh_trace.hip — wavefront BVH traversal on gfx950 (bvh_hit / bvh_hit_with_normal, utils/helperDi.slang:136-395).
//
// Result contract (bit-exact against the reference algorithm under the DESIGN.md FP policy):
//   the reference pops a node, slab-tests ITS box against [t_min, closest] (reject when t_max <= t_min, :149-170),
//   pushes left then right (right is visited first, no near/far ordering, :245-246) and, at a leaf, accepts any
//   Moller-Trumbore line hit regardless of t (:172-195); closest = min(t, closest).
// MI355X redesign with identical results:
//   * a node visit is ONE 64-byte record carrying both children's boxes (engine.hpp WideNode), so the slab
//     interval [tn, tf] of a child is computed when its parent is visited; the part of the reference test that
//     depends on `closest` (closest > tn) is re-evaluated when the deferred (left) child is popped — the same
//     accept/reject decision the reference takes at pop time;
//   * the right child is consumed immediately (no stack traffic), only left children are deferred;
//   * deferred entries live in an LDS-resident per-lane short stack (8 B x MR_LDS_STACK entries per lane,
//     bank-conflict free: lane-major), spilling to scratch only beyond that depth (never observed > 24);
//   * any-hit rays (shadow rays) leave at the first accepted triangle: `closest` cannot change before the first
//     hit, so the set of boxes that pass is order independent and the boolean equals the reference's exhaustive
//     search bit for bit;
//   * rays come from compacted queues; blocks grid-stride over the queue, count read on device (no host sync).
#include "engine.hpp"
#include "device_math.hpp"

namespace mr {

#define MR_STACK 64        // MAX_STACK_SIZE helperDi.slang:136
#define MR_LDS_STACK 12    // entries per lane kept in LDS before spilling to scratch
#define MR_TRACE_BLOCK 256

struct Slab { float tn, tf; };

// aabb_hit (helperDi.slang:149-170) split into its closest-independent part.
MR_DEV Slab slab(const float* __restrict__ bmin, const float* __restrict__ bmax, const float o[3], const float inv[3], float t_min) {
    float tn = t_min, tf = INFINITY;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float t0 = (bmin[i] - o[i]) * inv[i];
        float t1 = (bmax[i] - o[i]) * inv[i];
        if (inv[i] < 0.0f) { float tmp = t1; t1 = t0; t0 = tmp; }
        tn = t0 > tn ? t0 : tn;
        tf = t1 < tf ? t1 : tf;
    }
    Slab s; s.tn = tn; s.tf = tf; return s;
}

struct TraceOut { bool hit; float t, u, v; int slot; v3 d; };

template <bool ANY, bool COUNT>
MR_DEV TraceOut traverse(const BvhView& B, v3 ro, v3 rd_in, float t_min, float t_max, uint2* lds_stack, uint32_t* cnt) {
    TraceOut out; out.hit = false; out.t = 0.f; out.u = 0.f; out.v = 0.f; out.slot = -1;
    const v3 d = normalize(rd_in);  // helperDi.slang:201
    out.d = d;
    const float o[3] = {ro.x, ro.y, ro.z};
    float inv[3];
    {
        float dd[3] = {d.x, d.y, d.z};
#pragma unroll
        for (int i = 0; i < 3; i++) { float di = dd[i]; if (di == 0.f) di = 0.000001f; inv[i] = 1.0f / di; }
    }
    float closest = t_max;
    uint2 spill[MR_STACK - MR_LDS_STACK];
    int sp = 0;
    uint32_t popped = 1, entered = 0, leaves = 0, overflow = 0;
    // root (node 0) box test
    int cur;  // >= 0: internal node to visit; otherwise leaf (~slot) pending a triangle test; INT_MIN/2 = nothing
    const int NONE = 0x40000000;
    {
        Slab s = slab(B.root_box, B.root_box + 3, o, inv, t_min);
        cur = (s.tf > s.tn && closest > s.tn) ? 0 : NONE;
    }
    while (true) {
        if (cur == NONE) {
            // pop deferred (left) children until one passes the closest-dependent half of its box test
            bool found = false;
            while (sp > 0) {
                --sp;
                uint2 e = (sp < MR_LDS_STACK) ? lds_stack[sp * MR_TRACE_BLOCK] : spill[sp - MR_LDS_STACK];
                if (closest > __uint_as_float(e.y)) { cur = (int)e.x; found = true; break; }
            }
            if (!found) break;
        }
        if (cur >= 0) {
            if (COUNT) { entered++; popped += 2; }
            const WideNode* __restrict__ n = B.nodes + cur;
            // one 64-byte fetch: 4 x dwordx4
            const float4 q0 = reinterpret_cast<const float4*>(n)[0];
            const float4 q1 = reinterpret_cast<const float4*>(n)[1];
            const float4 q2 = reinterpret_cast<const float4*>(n)[2];
            const float4 q3 = reinterpret_cast<const float4*>(n)[3];
            const float lmin[3] = {q0.x, q0.y, q0.z}, lmax[3] = {q0.w, q1.x, q1.y};
            const float rmin[3] = {q1.z, q1.w, q2.x}, rmax[3] = {q2.y, q2.z, q2.w};
            const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
            Slab sl = slab(lmin, lmax, o, inv, t_min);
            Slab sr = slab(rmin, rmax, o, inv, t_min);
            if (sl.tf > sl.tn) {  // may still be rejected at pop time when closest has shrunk
                uint2 e; e.x = (uint32_t)left; e.y = __float_as_uint(sl.tn);
                if (sp < MR_LDS_STACK) lds_stack[sp * MR_TRACE_BLOCK] = e;
                else if (sp < MR_STACK) spill[sp - MR_LDS_STACK] = e;
                if (sp < MR_STACK) sp++; else overflow++;
            }
            cur = (sr.tf > sr.tn && closest > sr.tn) ? right : NONE;
            continue;
        }
        // leaf: triangle_hit (helperDi.slang:172-195), accepts any t
        {
            const int slot = ~cur;
            cur = NONE;
            if (COUNT) leaves++;
            const TriRec* __restrict__ tr = B.tris + slot;
            const float4 a = reinterpret_cast<const float4*>(tr)[0];
            const float4 b = reinterpret_cast<const float4*>(tr)[1];
            const float4 c = reinterpret_cast<const float4*>(tr)[2];
            const v3 v0 = V3(a.x, a.y, a.z), E1 = V3(a.w, b.x, b.y), E2 = V3(b.z, b.w, c.x);
            const v3 P = cross(d, E2);
            const float det = dot(E1, P);
            if (det > -1e-15f && det < 1e-15f) continue;
            const float invDet = 1 / det;
            const v3 Tv = ro - v0;
            const float u = dot(Tv, P) * invDet;
            if (u < 0 || u > 1) continue;
            const v3 Q = cross(Tv, E1);
            const float v = dot(d, Q) * invDet;
            if (v < 0 || u + v > 1) continue;
            const float t = dot(E2, Q) * invDet;
            out.hit = true;
            if (ANY) break;
            closest = fminf(t, closest);
            if (t <= closest) { out.u = u; out.v = v; out.slot = slot; }
            out.t = closest;
        }
    }
    if (COUNT && cnt) { cnt[0] = popped; cnt[1] = entered; cnt[2] = leaves; cnt[3] = overflow; }
    return out;
}

// closest-hit epilogue: pos / normal exactly as bvh_hit_with_normal reports them (:277-310, :372-384)
MR_DEV void finish_closest(const BvhView& B, const TraceOut& r, v3 ro, v3& pos, v3& nrm, int& prim) {
    pos = V3(0.f); nrm = V3(1.f); prim = -1;
    if (!r.hit) return;
    pos = ro + r.t * r.d;
    if (r.slot >= 0) {
        const TriRec* tr = B.tris + r.slot;
        v3 E1 = V3(tr->e1[0], tr->e1[1], tr->e1[2]), E2 = V3(tr->e2[0], tr->e2[1], tr->e2[2]);
        v3 fn = normalize(cross(E1, E2));
        float w = 1.0f - r.u - r.v;
        v3 n = r.u * fn + r.v * fn + w * fn;
        if (dot(-r.d, n) < 0) n = -n;
        nrm = normalize(n);
        prim = tr->prim;
    }
}

template <bool COUNT>
__global__ void __launch_bounds__(MR_TRACE_BLOCK) k_trace_any(BvhView B, const Ray* __restrict__ rays, const uint32_t* __restrict__ d_count,
                                                              uint32_t n_fixed, int32_t* __restrict__ hit, uint32_t* __restrict__ counters,
                                                              unsigned long long* __restrict__ stats) {
    __shared__ uint2 lds[MR_LDS_STACK * MR_TRACE_BLOCK];
    const uint32_t n = d_count ? *d_count : n_fixed;
    unsigned long long sp = 0, se = 0, sl = 0;
    for (uint32_t i = blockIdx.x * MR_TRACE_BLOCK + threadIdx.x; i < n; i += gridDim.x * MR_TRACE_BLOCK) {
        const float4 a = reinterpret_cast<const float4*>(rays + i)[0], b = reinterpret_cast<const float4*>(rays + i)[1];
        uint32_t c[4];
        TraceOut r = traverse<true, COUNT>(B, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), a.w, b.w, lds + threadIdx.x, c);
        hit[i] = r.hit ? 1 : 0;
        if (COUNT) {
            if (counters) { counters[4 * (size_t)i] = c[0]; counters[4 * (size_t)i + 1] = c[1]; counters[4 * (size_t)i + 2] = c[2]; counters[4 * (size_t)i + 3] = c[3]; }
            sp += c[0]; se += c[1]; sl += c[2];
        }
    }
    if (COUNT && stats) { atomicAdd(&stats[2], sp); atomicAdd(&stats[3], se); atomicAdd(&stats[4], sl); }
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[0], (unsigned long long)n);
}

template <bool COUNT>
__global__ void __launch_bounds__(MR_TRACE_BLOCK) k_trace_closest(BvhView B, const Ray* __restrict__ rays, const uint32_t* __restrict__ d_count,
                                                                  uint32_t n_fixed, HitRec* __restrict__ rec, int32_t* __restrict__ hit,
                                                                  float* __restrict__ t, float* __restrict__ pos, float* __restrict__ normal,
                                                                  int32_t* __restrict__ prim, uint32_t* __restrict__ counters,
                                                                  unsigned long long* __restrict__ stats) {
    __shared__ uint2 lds[MR_LDS_STACK * MR_TRACE_BLOCK];
    const uint32_t n = d_count ? *d_count : n_fixed;
    unsigned long long sp = 0, se = 0, sl = 0;
    for (uint32_t i = blockIdx.x * MR_TRACE_BLOCK + threadIdx.x; i < n; i += gridDim.x * MR_TRACE_BLOCK) {
        const float4 a = reinterpret_cast<const float4*>(rays + i)[0], b = reinterpret_cast<const float4*>(rays + i)[1];
        uint32_t c[4];
        const v3 ro = V3(a.x, a.y, a.z);
        TraceOut r = traverse<false, COUNT>(B, ro, V3(b.x, b.y, b.z), a.w, b.w, lds + threadIdx.x, c);
        v3 p, nn; int pr;
        finish_closest(B, r, ro, p, nn, pr);
        if (rec) {
            float4 o0, o1;
            o0.x = p.x; o0.y = p.y; o0.z = p.z; o0.w = __int_as_float(r.hit ? 1 : 0);
            o1.x = nn.x; o1.y = nn.y; o1.z = nn.z; o1.w = r.t;
            reinterpret_cast<float4*>(rec + i)[0] = o0; reinterpret_cast<float4*>(rec + i)[1] = o1;
        }
        if (hit) hit[i] = r.hit ? 1 : 0;
        if (t) t[i] = r.t;
        if (pos) st3(pos, i, p);
        if (normal) st3(normal, i, nn);
        if (prim) prim[i] = pr;
        if (COUNT) {
            if (counters) { counters[4 * (size_t)i] = c[0]; counters[4 * (size_t)i + 1] = c[1]; counters[4 * (size_t)i + 2] = c[2]; counters[4 * (size_t)i + 3] = c[3]; }
            sp += c[0]; se += c[1]; sl += c[2];
        }
    }
    if (COUNT && stats) { atomicAdd(&stats[5], sp); atomicAdd(&stats[6], se); atomicAdd(&stats[7], sl); }
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[1], (unsigned long long)n);
}

static int trace_grid(size_t capacity) {
    // persistent-style launch: enough 256-thread blocks to fill 256 CUs several times over, grid-stride beyond
    size_t want = (capacity + MR_TRACE_BLOCK - 1) / MR_TRACE_BLOCK;
    size_t cap = 256 * 8;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

int trace_any_queue(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit,
                    unsigned long long* stats, hipStream_t s) {
    k_trace_any<false><<<trace_grid(capacity), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, hit, nullptr, stats);
    MR_LAUNCH_CHECK("trace_any_queue");
    return 0;
}
int trace_closest_queue(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, HitRec* out,
                        unsigned long long* stats, hipStream_t s) {
    k_trace_closest<false><<<trace_grid(capacity), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, out, nullptr, nullptr,
                                                                            nullptr, nullptr, nullptr, nullptr, stats);
    MR_LAUNCH_CHECK("trace_closest_queue");
    return 0;
}
int trace_any_queue_counted(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit,
                            unsigned long long* stats, hipStream_t s) {
    k_trace_any<true><<<trace_grid(capacity), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, hit, nullptr, stats);
    MR_LAUNCH_CHECK("trace_any_queue_counted");
    return 0;
}
int trace_closest_queue_counted(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, HitRec* out,
                                unsigned long long* stats, hipStream_t s) {
    k_trace_closest<true><<<trace_grid(capacity), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, out, nullptr, nullptr,
                                                                           nullptr, nullptr, nullptr, nullptr, stats);
    MR_LAUNCH_CHECK("trace_closest_queue_counted");
    return 0;
}

}  // namespace mr

using namespace mr;

extern "C" int mirres_bvh_trace(mirres_bvh_t* bvh, const float* rays, int n, int mode, int32_t* hit, float* t, float* pos, float* normal,
                                int32_t* prim, uint32_t* counters, void* stream) {
    if (!bvh || !rays || n < 0 || (mode != 0 && mode != 1)) { set_error("mirres_bvh_trace: bad argument"); return MIRRES_E_ARG; }
    if (bvh->T < 2) { set_error("mirres_bvh_trace: BVH not built"); return MIRRES_E_STATE; }
    if (n == 0) return MIRRES_OK;
    hipStream_t s = (hipStream_t)stream;
    const Ray* r = reinterpret_cast<const Ray*>(rays);
    const int g = trace_grid((size_t)n);
    if (mode == 0) {
        if (!hit) { set_error("mirres_bvh_trace: any-hit needs hit[]"); return MIRRES_E_ARG; }
        if (counters) k_trace_any<true><<<g, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), r, nullptr, (uint32_t)n, hit, counters, nullptr);
        else k_trace_any<false><<<g, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), r, nullptr, (uint32_t)n, hit, nullptr, nullptr);
    } else {
        if (counters) k_trace_closest<true><<<g, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), r, nullptr, (uint32_t)n, nullptr, hit, t, pos, normal, prim, counters, nullptr);
        else k_trace_closest<false><<<g, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), r, nullptr, (uint32_t)n, nullptr, hit, t, pos, normal, prim, nullptr, nullptr);
    }
    MR_LAUNCH_CHECK("mirres_bvh_trace");
    return MIRRES_OK;
}
