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
#include <cstdlib>

namespace mr {

#define MR_STACK 64        // MAX_STACK_SIZE helperDi.slang:136
#ifndef MR_LDS_STACK
#define MR_LDS_STACK 12    // entries per lane kept in LDS before spilling to scratch
#endif
#ifndef MR_TRACE_BLOCK
#define MR_TRACE_BLOCK 64  // one wave per workgroup: nothing in the traversal kernels is shared between the waves of a workgroup any more (the LDS stack is per lane; the staged top
                           // levels are off by default), and single-wave workgroups are placed and retired independently: lego-like frame +1.0 %, icosphere +- 0 against 256
                           // (rounds 1-4), the same number of waves launched (profiles/r04_ab_trace_block.txt)
#endif

struct Slab { float tn, tf; };

// aabb_hit (helperDi.slang:149-170) split into its closest-independent part. The reference computes t0/t1 per axis, swaps them when
// inv < 0 and folds them with `t0 > t_min ? t0 : t_min` / `t1 < t_max ? t1 : t_max`, rejecting when t_max <= t_min. The folds are max / min that
// IGNORE a NaN plane (the comparison is false), which is what fmaxf / fminf do -> v_max3 / v_min3; the swap is kept as a select on the sign
// of the reciprocal rather than min / max of the two products: for finite products the two are the same ((b - o) * inv is monotone in b),
// but a NaN product — 0 * inf: the origin exactly on a box plane and an infinite reciprocal (a denormal direction component) — must stay in
// ITS slot to be ignored there, as in the reference; min / max would replace it by the other plane. Same instruction count either way.
MR_DEV Slab slab(const float* __restrict__ bmin, const float* __restrict__ bmax, const float o[3], const float inv[3], float t_min) {
    const float ax = (bmin[0] - o[0]) * inv[0], bx = (bmax[0] - o[0]) * inv[0];
    const float ay = (bmin[1] - o[1]) * inv[1], by = (bmax[1] - o[1]) * inv[1];
    const float az = (bmin[2] - o[2]) * inv[2], bz = (bmax[2] - o[2]) * inv[2];
    const bool sx = inv[0] < 0.f, sy = inv[1] < 0.f, sz = inv[2] < 0.f;
    Slab s;
    s.tn = fmaxf(fmaxf(fmaxf(sx ? bx : ax, sy ? by : ay), sz ? bz : az), t_min);
    s.tf = fminf(fminf(sx ? ax : bx, sy ? ay : by), sz ? az : bz);
    return s;
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


// ---------------------------------------------------------------- persistent "while-while" traversal with lane refill
// PMC on the one-ray-per-thread kernels above: VALU lane utilisation 24 %, 60 % of wave cycles in s_waitcnt — a wave lives as long
// as its longest ray. Here a wave owns a chunk of the queue (one atomic per MR_CHUNK rays) and, whenever fewer than MR_REFILL lanes
// are still traversing, idle lanes pull the next rays of the chunk with pure ballot/popcount arithmetic (no atomics). Per-ray
// arithmetic and visit order are exactly those of traverse<>, so results are bit-identical to the simple kernels.
#define MR_CHUNK_MAX 1024
#ifndef MR_CHUNK_DIV
#define MR_CHUNK_DIV 4      // chunks per launched wave: a chunk is n / (waves x MR_CHUNK_DIV) rays, at least 64
#endif
#ifndef MR_REFILL
#define MR_REFILL 40
#endif
// Work distribution. A single queue-head word serialises at ~88 returning atomics/us on MI355X: with 64-ray chunks a 2.3 M-ray launch needs
// 36 k of them = 0.41 ms — the whole kernel. The ray queue is therefore cut into MR_NQ contiguous sub-queues with their own head words
// (128 B apart: different L2 channels); a wave starts on sub-queue (wave id mod MR_NQ), takes chunks from it and moves to the next one when
// it runs dry (each sub-queue is probed at most once after it emptied, so a wave stops after MR_NQ failed probes).
// The END of a launch (round 5). Every wave leaves through MR_NQ failed probes; as returning atomicAdds those are 8192 waves x 32 = 262 k read-modify-writes,
// 8192 per word, ~11 ns each when they queue on one word: an EMPTY launch takes 108 us doing nothing else (profiles/r05_strip_fixed_cost.txt), and a strip of an
// 8-GPU frame spends more time there than on its rays. What does NOT help, measured (profiles/r05_ab_grab_modes.txt): probing with a load of the head first
// (0.67 -> 0.80 ms on 6.9 M rays), or every failing wave OR-ing "empty" into a shared mask word (1.10 ms) — same instruction counts, but TCP_TCR_TCP_STALL_CYCLES
// 1.3 M -> 187 M and twice the L1 miss latency: a few hundred thousand atomics queued on ONE line (11 ns each, serial) saturate that line's L2 channel for the whole
// launch, and every node fetch that maps to the channel waits behind them (profiles/r05_pmc_grab_modes.txt). The failed probes have to go, not get cheaper:
// the wave whose atomicAdd lands in [end, end + chunk) — exactly one per sub-queue, chunks being equal — publishes the sub-queue's bit in a mask word on a line of
// its own (32 atomics per launch), and a wave that has just failed reads the mask once and skips what is known to be empty. Sub-queues beyond n are never touched.
// Which wave gets which rays does not change any ray's answer; a mask that lags only costs the atomic it would have saved.
#ifndef MR_GRAB_MODE
#define MR_GRAB_MODE 3     // 0: every probe is an atomicAdd (rounds 1-4); 3: + the mask of empty sub-queues, published once per sub-queue
#endif
MR_DEV bool grab_chunk(uint32_t* __restrict__ heads, uint32_t n, uint32_t per, uint32_t chunk, uint32_t& q, uint32_t& fails, uint32_t& known, uint32_t& c_next, uint32_t& c_end, int lane) {
    uint32_t* const empty_mask = heads + MR_NQ * MR_QSTRIDE;
    while (fails < MR_NQ) {
        const uint32_t qb = q * per;
        if (MR_GRAB_MODE == 0 || (qb < n && !((known >> q) & 1u))) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(heads + q * MR_QSTRIDE, chunk);
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t qe = (qb + per < n) ? qb + per : n;
            if (qb < n && base < qe - qb) { c_next = qb + base; c_end = (c_next + chunk < qe) ? c_next + chunk : qe; return true; }
            if (MR_GRAB_MODE == 3) {
                uint32_t m = 0;
                if (lane == 0) {
                    if (base - (qe - qb) < chunk) (void)__hip_atomic_fetch_or(empty_mask, 1u << q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    m = __hip_atomic_load(empty_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                known |= __builtin_amdgcn_readfirstlane(m) | (1u << q);
            }
        }
        q = (q + 1 == MR_NQ) ? 0 : q + 1; fails++;
    }
    return false;
}

template <bool ANY, bool FRONT = false>   // FRONT: a conventional closest hit — only triangles met in (t_min, t_max] count (mirres_bvh_trace mode 4; the reference's own rule is FRONT = false)
__global__ void __launch_bounds__(MR_TRACE_BLOCK) k_trace_persist(BvhView B, const Ray* __restrict__ rays, const uint32_t* __restrict__ d_count,
                                                                  uint32_t n_fixed, uint32_t* __restrict__ work_head, int32_t* __restrict__ hit_out,
                                                                  HitRec* __restrict__ rec, float* __restrict__ t_out, float* __restrict__ pos_out,
                                                                  float* __restrict__ nrm_out, int32_t* __restrict__ prim_out,
                                                                  unsigned long long* __restrict__ stats, const uint32_t* __restrict__ redo = nullptr) {
    __shared__ uint2 lds[MR_LDS_STACK * MR_TRACE_BLOCK];
    uint2* const lds_stack = lds + threadIdx.x;
    const uint32_t n = d_count ? *d_count : n_fixed;
    const int lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int NONE = 0x40000000;
    // chunk size: ~4 chunks per resident wave so that the tail balances, 64..1024 rays (one global atomic per chunk)
    uint32_t chunk = n / (gridDim.x * (MR_TRACE_BLOCK / 64) * MR_CHUNK_DIV);
    chunk = chunk < 64 ? 64 : (chunk > MR_CHUNK_MAX ? MR_CHUNK_MAX : (chunk & ~63u));
    const uint32_t q_per = (((n + MR_NQ - 1) / MR_NQ) + 63u) & ~63u;                                  // rays per sub-queue
    uint32_t q_cur = (blockIdx.x * (MR_TRACE_BLOCK / 64) + (threadIdx.x >> 6)) % MR_NQ, q_fail = 0;   // wave-uniform
    uint32_t q_known = 0;   // sub-queues this wave knows to be empty (grab_chunk)
    uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform
    bool exhausted = false;                  // wave-uniform: the global queue has no more chunks
    bool have = false;
    // per-lane ray state
    float o[3], inv[3]; v3 d = V3(0.f), ro = V3(0.f);
    float t_min = 0.f, closest = 0.f, best_t = 0.f, best_u = 0.f, best_v = 0.f;
    int cur = NONE, sp = 0, best_slot = -1; uint32_t ridx = 0; bool any_hit = false;
    uint2 spill[MR_STACK - MR_LDS_STACK];
    while (true) {
        // ---- refill idle lanes from the wave's chunk
        const uint64_t need = __ballot(!have);
        if (need && !exhausted) {
            if (chunk_next >= chunk_end) exhausted = !grab_chunk(work_head, n, q_per, chunk, q_cur, q_fail, q_known, chunk_next, chunk_end, lane);
            if (!exhausted) {
                const uint32_t idx = chunk_next + (uint32_t)__popcll(need & lt_mask);
                if (!have && idx < chunk_end) {
                    const uint32_t rid = redo ? redo[idx] : idx;   // redo: the few rays the ordered fast path hands back (see k_trace_closest4)
                    const float4 a = reinterpret_cast<const float4*>(rays + rid)[0], b = reinterpret_cast<const float4*>(rays + rid)[1];
                    ridx = rid; ro = V3(a.x, a.y, a.z); t_min = a.w; closest = b.w;
                    d = normalize(V3(b.x, b.y, b.z));
                    o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
                    { float dd[3] = {d.x, d.y, d.z};
#pragma unroll
                      for (int i = 0; i < 3; i++) { float di = dd[i]; if (di == 0.f) di = 0.000001f; inv[i] = 1.0f / di; } }
                    sp = 0; any_hit = false; best_t = 0.f; best_u = 0.f; best_v = 0.f; best_slot = -1;
                    Slab s0 = slab(B.root_box, B.root_box + 3, o, inv, t_min);
                    cur = (s0.tf > s0.tn && closest > s0.tn) ? 0 : NONE;
                    have = true;
                }
                const uint32_t want = (uint32_t)__popcll(need);
                chunk_next = (chunk_next + want < chunk_end) ? chunk_next + want : chunk_end;
            }
        }
        if (!__ballot(have)) { if (exhausted) break; else continue; }
        // ---- traverse until too few lanes are busy
        do {
            if (have) {
                bool done = false;
                if (cur == NONE) {
                    bool found = false;
                    while (sp > 0) {
                        --sp;
                        uint2 e = (sp < MR_LDS_STACK) ? lds_stack[sp * MR_TRACE_BLOCK] : spill[sp - MR_LDS_STACK];
                        if (closest > __uint_as_float(e.y)) { cur = (int)e.x; found = true; break; }
                    }
                    if (!found) done = true;
                }
                if (!done) {
                    if (cur >= 0) {
                        const WideNode* __restrict__ nd = B.nodes + cur;
                        const float4 q0 = reinterpret_cast<const float4*>(nd)[0];
                        const float4 q1 = reinterpret_cast<const float4*>(nd)[1];
                        const float4 q2 = reinterpret_cast<const float4*>(nd)[2];
                        const float4 q3 = reinterpret_cast<const float4*>(nd)[3];
                        const float lmin[3] = {q0.x, q0.y, q0.z}, lmax[3] = {q0.w, q1.x, q1.y};
                        const float rmin[3] = {q1.z, q1.w, q2.x}, rmax[3] = {q2.y, q2.z, q2.w};
                        const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
                        Slab sl = slab(lmin, lmax, o, inv, t_min);
                        Slab sr = slab(rmin, rmax, o, inv, t_min);
                        if (sl.tf > sl.tn) {
                            uint2 e; e.x = (uint32_t)left; e.y = __float_as_uint(sl.tn);
                            if (sp < MR_LDS_STACK) lds_stack[sp * MR_TRACE_BLOCK] = e;
                            else if (sp < MR_STACK) spill[sp - MR_LDS_STACK] = e;
                            if (sp < MR_STACK) sp++;
                        }
                        cur = (sr.tf > sr.tn && closest > sr.tn) ? right : NONE;
                    } else {
                        const int slot = ~cur;
                        cur = NONE;
                        const TriRec* __restrict__ tr = B.tris + slot;
                        const float4 a = reinterpret_cast<const float4*>(tr)[0];
                        const float4 b = reinterpret_cast<const float4*>(tr)[1];
                        const float4 c = reinterpret_cast<const float4*>(tr)[2];
                        const v3 v0 = V3(a.x, a.y, a.z), E1 = V3(a.w, b.x, b.y), E2 = V3(b.z, b.w, c.x);
                        const v3 P = cross(d, E2);
                        const float det = dot(E1, P);
                        if (!(det > -1e-15f && det < 1e-15f)) {
                            const float invDet = 1 / det;
                            const v3 Tv = ro - v0;
                            const float u = dot(Tv, P) * invDet;
                            if (!(u < 0 || u > 1)) {
                                const v3 Q = cross(Tv, E1);
                                const float v = dot(d, Q) * invDet;
                                if (!(v < 0 || u + v > 1)) {
                                    if (ANY) { any_hit = true; done = true; }
                                    else {
                                        const float t = dot(E2, Q) * invDet;
                                        if (!FRONT || (t > t_min && t <= closest)) {
                                            any_hit = true;
                                            closest = fminf(t, closest);
                                            if (t <= closest) { best_u = u; best_v = v; best_slot = slot; }
                                            best_t = closest;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                if (done) {
                    have = false;
                    if (ANY) hit_out[ridx] = any_hit ? 1 : 0;
                    else {
                        TraceOut r; r.hit = any_hit; r.t = best_t; r.u = best_u; r.v = best_v; r.slot = best_slot; r.d = d;
                        v3 p, nn; int pr;
                        finish_closest(B, r, ro, p, nn, pr);
                        if (rec) {
                            float4 o0, o1;
                            o0.x = p.x; o0.y = p.y; o0.z = p.z; o0.w = __int_as_float(any_hit ? 1 : 0);
                            o1.x = nn.x; o1.y = nn.y; o1.z = nn.z; o1.w = best_t;
                            reinterpret_cast<float4*>(rec + ridx)[0] = o0; reinterpret_cast<float4*>(rec + ridx)[1] = o1;
                        }
                        if (hit_out) hit_out[ridx] = any_hit ? 1 : 0;
                        if (t_out) t_out[ridx] = best_t;
                        if (pos_out) st3(pos_out, ridx, p);
                        if (nrm_out) st3(nrm_out, ridx, nn);
                        if (prim_out) prim_out[ridx] = pr;
                    }
                }
            }
        } while (__popcll(__ballot(have)) >= MR_REFILL || (exhausted && __ballot(have)));
    }
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[redo ? 10 : (ANY ? 0 : 1)], (unsigned long long)n);   // [10]: rays the ordered fast path handed back
}


// ---------------------------------------------------------------- shadow rays: order-free any-hit traversal
// For an any-hit query `closest` stays t_max until the first accepted triangle, so (1) the reference's result is the OR over all leaves
// whose OWN box passes the slab test of "triangle_hit accepts": a leaf's ancestors always pass when the leaf does (their boxes are
// fmin/fmax unions and (b - o) * inv is monotone in b under IEEE rounding), hence (2) any visiting order and any pruning by ancestor
// boxes yields the same boolean. The stack holds bare 4-byte references in LDS (16 per lane, scratch beyond); no pop-time re-test is needed.
#ifndef MR_ANY_LDS
#define MR_ANY_LDS 16
#endif
// Depth of the shadow-ray kernel's private stack (LDS part + scratch). A 4-wide node defers at most three references and the 4-wide collapse is never
// deeper than the LBVH it was collapsed from, whose depth is bounded by the bits of the augmented sort key: 30 Morton bits + ceil(log2 T) position bits
// (lbvh_hierarchy.slang:40-60: every internal node splits its range at the highest differing bit of (code, position)). Hence at most
// 3 * (30 + ceil(log2 T)) entries for the collapsed reference LBVH; the private extended-Morton hierarchy (bvh_build.hip) has 32 + 6 key bits that can differ:
// 3 * (38 + ceil(log2 T)); with its upper levels rebuilt by the SAH top (at most 40 levels above clusters that share 24 key bits: 14 key bits + the position
// bits left below) 3 * (40 + 14 + ceil(log2 T)) = 219 for T < 2^19, 255 for any T < 2^31 — MR_ANY_STACK = 256 cannot overflow ("Stack bounds").
#ifndef MR_ANY_STACK
#define MR_ANY_STACK 256
#endif
// 3 deferred references per 4-wide level x (levels of the SAH top + key bits free below a cluster + position bits of an int32 triangle count)
static_assert(MR_ANY_STACK >= 3 * (MR_SAH_LEVELS + (38 - MR_SAH_PREFIX) + 31), "MR_ANY_STACK must cover the deepest private hierarchy (DESIGN.md, stack bounds)");
static_assert(MR_ANY_LDS >= 3 && MR_ANY_LDS <= MR_ANY_STACK, "the LDS part of the private stack holds at least one node's deferred references");
#define MR_TOPBIT 0x20000000
#ifndef MR_ANY_SEL
#define MR_ANY_SEL 1      // straight-line child selection (round 6); 0: round 5's sequential insert
#endif
#ifndef MR_ANY_POP
#define MR_ANY_POP 1      // the pop as straight-line code (round 6: frames +0.3 … +2 % / +0.3 … +0.7 %, profiles/r06_ab_any_pop.txt); 0: round 5's nested ifs
#endif
#ifndef MR_CL_SEL
#define MR_CL_SEL 1      // unconditional pushes in the ordered closest-hit kernel (round 6: kernel -5 %, frames +0.6 % / +1.1 %, profiles/r06_ab_closest_sel.txt); 0: round 5's exec-masked pushes
#endif
#ifndef MR_ANY_LEANREFILL
#define MR_ANY_LEANREFILL 1      // short division / square root where the pixel-pair source forms its rays (round 6); 0: the compiler's IEEE sequences
#endif
#ifndef MR_CL_REFILL
#define MR_CL_REFILL MR_REFILL
#endif
// ---------------------------------------------------------------- shadow rays on the COMPRESSED 4-wide collapse (engine.hpp Node4q / LeafRec)
// The any-hit bit is the OR over leaves whose own box passes the slab test (above); interior boxes only steer the search and may be any
// supersets. Node4q stores them as 8-bit outward-rounded offsets (64 B per visit = 4 dwordx4 gathers instead of 8 — the kernel is bound by
// vector-L1 lookups of divergent gathers); (b' - o) * inv is monotone in b', so a decoded box passes whenever the exact one does and no
// leaf the reference visits is missed. A leaf that passes the conservative test is re-tested against its exact LBVH box (LeafRec) before
// the triangle test, so the set of triangles tested — and therefore the result — is the reference's, bit for bit.
// ---------------------------------------------------------------- interior boxes of the compressed tree: one fma per plane
// Interior boxes only steer the search: any test that passes whenever the exact slab test of a descendant leaf's own box passes is admissible
// (see above). The decoded-plane test  t = (fmaf(q, s, g) - o) * inv  costs cvt + fma + sub + mul per plane; algebraically it is
//   t = q * (s * inv) + (g - o) * inv,
// i.e. ONE fma per plane once  S = s * inv  (exact: s is a power of two) and  B = (g - o) * inv  are formed per node and axis. The two
// expressions round differently, so the fused one is made conservative by a margin m per axis, subtracted from the entry and added to the exit
// distances. With u = 2^-24, D = q s + g the decoded plane (|D| <= Bs, the largest coordinate magnitude of the scene box) and R = (D - o) inv:
//   decoded form:  |t_dec - R|  <=  u |inv| (|D| + 2 |D - o|)            <= 3 u |inv| (Bs + |o|)
//   fused form:    |t_fus -+ m - R| <= u |inv| (2 |g - o| + 2 |D - o|) + 2 u m  <= 4 u |inv| (Bs + |o|) + 2 u m
// so m = 16 u |inv| (Bs + |o|) = 2^-20 |inv| (Bs + |o|) puts the fused entry distance at or below, and the fused exit distance at or above, the
// decoded ones — which in turn bound every descendant leaf's exact slab values (monotonicity of (b - o) * inv in b, DESIGN.md §Traversal
// exactness). The margin is ~1e-6 of the scene size in t: nothing against the 8-bit quantisation of the boxes. Directions with an infinite
// reciprocal (a denormal component) yield inf / NaN planes on that axis; fmaxf / fminf drop NaNs, +-inf margins only widen, and a normalised
// direction has at least one finite reciprocal, so such a ray is tested on its remaining axes: permissive, never wrong.
struct RayCons { float mx, my, mz; };
MR_DEV float scene_bound(const float* __restrict__ root_box) {
    float b = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) b = fmaxf(b, fabsf(root_box[i]));
    return b;
}
MR_DEV RayCons ray_margins(float bs, float ox, float oy, float oz, float ix, float iy, float iz) {
    RayCons c; const float k = 9.5367431640625e-07f;   // 2^-20
    c.mx = k * fabsf(ix) * (bs + fabsf(ox)); c.my = k * fabsf(iy) * (bs + fabsf(oy)); c.mz = k * fabsf(iz) * (bs + fabsf(oz));
    return c;
}
struct NodeCons { float Sx, Sy, Sz, Bnx, Bny, Bnz, Bfx, Bfy, Bfz; uint32_t nqx, fqx, nqy, fqy, nqz, fqz; };
MR_DEV NodeCons node_cons(const uint4& h0, const uint4& h1, const uint4& h2, float ox, float oy, float oz, float ix, float iy, float iz, const RayCons& m) {
    NodeCons c;
    const float gx = __uint_as_float(h0.x), gy = __uint_as_float(h0.y), gz = __uint_as_float(h0.z);
    const float sx = __uint_as_float(h0.w), sy = __uint_as_float(h2.z), sz = __uint_as_float(h2.w);   // Node4q::step_x/y/z
    // near / far plane of each axis picked by the sign of the direction (nqx = the byte word holding the planes the ray meets first); an unused
    // entry has lo = 255 > hi = 0 on every axis (entry beyond exit by the whole node extent) and refers to the null leaf
    c.nqx = ix >= 0.f ? h1.x : h1.w; c.fqx = ix >= 0.f ? h1.w : h1.x;
    c.nqy = iy >= 0.f ? h1.y : h2.x; c.fqy = iy >= 0.f ? h2.x : h1.y;
    c.nqz = iz >= 0.f ? h1.z : h2.y; c.fqz = iz >= 0.f ? h2.y : h1.z;
    c.Sx = sx * ix; c.Sy = sy * iy; c.Sz = sz * iz;
    const float bx = (gx - ox) * ix, by = (gy - oy) * iy, bz = (gz - oz) * iz;
    c.Bnx = bx - m.mx; c.Bny = by - m.my; c.Bnz = bz - m.mz;
    c.Bfx = bx + m.mx; c.Bfy = by + m.my; c.Bfz = bz + m.mz;
    return c;
}
// entry / exit distance of child k (compile-time k: the byte selects fold into v_cvt_f32_ubyteK)
template <int K>
MR_DEV void child_slab(const NodeCons& c, float t_min, float& tn, float& tf) {
    const float nx = fmaf((float)((c.nqx >> (8 * K)) & 0xffu), c.Sx, c.Bnx), fx_ = fmaf((float)((c.fqx >> (8 * K)) & 0xffu), c.Sx, c.Bfx);
    const float ny = fmaf((float)((c.nqy >> (8 * K)) & 0xffu), c.Sy, c.Bny), fy_ = fmaf((float)((c.fqy >> (8 * K)) & 0xffu), c.Sy, c.Bfy);
    const float nz = fmaf((float)((c.nqz >> (8 * K)) & 0xffu), c.Sz, c.Bnz), fz_ = fmaf((float)((c.fqz >> (8 * K)) & 0xffu), c.Sz, c.Bfz);
    tn = fmaxf(fmaxf(fmaxf(nx, ny), nz), t_min);
    tf = fminf(fminf(fx_, fy_), fz_);
}
MR_DEV void child_slabs(const NodeCons& c, float t_min, float tn[4], float tf[4]) {
    child_slab<0>(c, t_min, tn[0], tf[0]); child_slab<1>(c, t_min, tn[1], tf[1]); child_slab<2>(c, t_min, tn[2], tf[2]); child_slab<3>(c, t_min, tn[3], tf[3]);
}

// normalize() / oct_decode() with the short reciprocal and square root (device_math.hpp) for arguments known to be in their exact range: a squared length in [1/3, 3]
MR_DEV v3 normalize_lean(v3 v) {
    const float dd = dot(v, v);
    const float yr = __builtin_amdgcn_rsqf(dd); float sq = dd * yr;
    sq = __builtin_fmaf(__builtin_fmaf(-sq, sq, dd), 0.5f * yr, sq);          // mr_sqrt for a normal, non-zero argument
    return v * lean_rcp(sq);
}
MR_DEV v3 oct_decode_lean(v2 f) {     // oct_decode (device_math.hpp, helperDi.slang:123-134): |n.x| + |n.y| + |n.z| = 1, so the squared length lies in [1/3, 1]
    float fx = f.x * 2.0f - 1.0f, fy = f.y * 2.0f - 1.0f;
    v3 n = V3(fx, fy, (1.0f - fabsf(fx)) - fabsf(fy));
    float t = clampf(-n.z, 0.0f, 1.0f);
    n.x += (n.x >= 0.0f ? -t : t);
    n.y += (n.y >= 0.0f ? -t : t);
    return normalize_lean(n);
}
template <bool FRONT = false>   // FRONT: a conventional occlusion query — the hit must lie in front of the origin (t > 0); the reference's bvh_hit does not look at t
MR_DEV bool tri_accepts_regs(float4 a, float4 b, float4 c, v3 ro, v3 d) {
    const v3 v0 = V3(a.x, a.y, a.z), E1 = V3(a.w, b.x, b.y), E2 = V3(b.z, b.w, c.x);
    const v3 P = cross(d, E2);
    const float det = dot(E1, P);
    if (det > -1e-15f && det < 1e-15f) return false;
    // |det| >= 1e-15 here (or NaN / inf, which v_div_fixup_f32 treats as the compiler's division does): inside the range in which the short
    // reciprocal returns the bits of 1 / det (device_math.hpp; mirres_selfcheck_arith) unless |det| > 2^102, i.e. triangle edges beyond 2^51
    const float invDet = lean_rcp(det);
    const v3 Tv = ro - v0;
    const float u = dot(Tv, P) * invDet;
    if (u < 0 || u > 1) return false;
    const v3 Q = cross(Tv, E1);
    const float v = dot(d, Q) * invDet;
    if (v < 0 || u + v > 1) return false;
    if (FRONT) return dot(E2, Q) * invDet > 0.f;
    return true;
}
// TIMED = 2: front-only occlusion (nerf/render_dump.py's external `intersector`, a conventional ray tracer) — the same traversal with t > 0 required
// SRC = 1: the queue holds one pixel pair per two rays and the ray is formed here (engine.hpp RaySrc)
template <bool COUNT, int TOPN, int TIMED = 0, int SRC = 0>   // TIMED: identical code under a second name, launched by bench.py's event-timed frame so that a rocprofv3
                                                              // kernel trace of the same command shows those launches as their own row
__global__ void __launch_bounds__(MR_TRACE_BLOCK) k_trace_any4q(BvhView B, const Ray* __restrict__ rays, const uint32_t* __restrict__ d_count,
                                                               uint32_t n_fixed, uint32_t* __restrict__ work_head, int32_t* __restrict__ hit_out,
                                                               unsigned long long* __restrict__ stats, RaySrc src = RaySrc{nullptr, nullptr, 0.f, 0}) {
    __shared__ uint32_t lds[(MR_ANY_LDS + (MR_ANY_SEL ? 1 : 0)) * MR_TRACE_BLOCK];   // MR_ANY_SEL: one spare row for the last unconditional store of the branch-free child selection
    __shared__ __attribute__((aligned(16))) uint4 s_top[TOPN > 0 ? TOPN * 4 : 1];
    if (TOPN > 0) {
        const uint4* src = reinterpret_cast<const uint4*>(TOPN > 85 ? B.top341q : B.top85q);
        for (int i = threadIdx.x; i < TOPN * 4; i += MR_TRACE_BLOCK) s_top[i] = src[i];
        __syncthreads();
    }
    if (B.dbg && (threadIdx.x & 63) == 0) B.dbg[2 * (blockIdx.x * (MR_TRACE_BLOCK / 64) + (threadIdx.x >> 6))] = wall_clock64();
    uint32_t* const lds_stack = lds + threadIdx.x;
    const uint32_t n = d_count ? *d_count : n_fixed;
    const int lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t chunk = n / (gridDim.x * (MR_TRACE_BLOCK / 64) * MR_CHUNK_DIV);
    chunk = chunk < 64 ? 64 : (chunk > MR_CHUNK_MAX ? MR_CHUNK_MAX : (chunk & ~63u));
    const uint32_t q_per = (((n + MR_NQ - 1) / MR_NQ) + 63u) & ~63u;                                  // rays per sub-queue
    uint32_t q_cur = (blockIdx.x * (MR_TRACE_BLOCK / 64) + (threadIdx.x >> 6)) % MR_NQ, q_fail = 0;   // wave-uniform
    uint32_t q_known = 0;   // sub-queues this wave knows to be empty (grab_chunk)
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false, have = false;
    float ox = 0, oy = 0, oz = 0, ix = 0, iy = 0, iz = 0; v3 d = V3(0.f), ro = V3(0.f);
    float t_min = 0.f, t_max = 0.f;
    RayCons rc; rc.mx = rc.my = rc.mz = 0.f;
    const float scene_bs = scene_bound(B.root_box);
    int cur = 0, sp = 0, sbase = 0; uint32_t ridx = 0;   // the lane's deferred entries live in [sbase, sp)
    uint32_t spill[MR_ANY_STACK - MR_ANY_LDS];
    unsigned long long c_boxes = 0, c_nodes = 0, c_leaves = 0; int c_maxsp = 0;
    unsigned long long w_iters = 0, w_leaf_iters = 0, w_leaf_lanes = 0;      // COUNT: wave iterations, those that ran the leaf branch, leaf visits (wave-uniform; lane 0 reports)
    while (true) {
        const uint64_t need = __ballot(!have);
        if (need && exhausted) {
            // ---- the tail of the launch: the queue is empty and the wave waits for its longest rays. A shadow ray's answer is an OR over subtrees, so
            // a lane with deferred entries hands its OLDEST one (the bottom of its stack: usually the largest subtree) to an idle lane, which searches
            // it for the same ray; whoever finds an occluder writes 1 (the slot was zeroed when the ray was loaded).
            const uint64_t donors = __ballot(have && sp > sbase && sbase < MR_ANY_LDS);
            if (donors) {
                const int nd = __popcll(donors), ni = __popcll(need);
                const int my_idle = __popcll(need & lt_mask), my_don = __popcll(donors & lt_mask);
                const bool take = !have && my_idle < nd;
                const bool give = have && sp > sbase && sbase < MR_ANY_LDS && my_don < ni;
                uint64_t m = donors;
                if (take) for (int t = 0; t < my_idle; t++) m &= m - 1;
                const int src = take ? __builtin_ctzll(m) : lane;
                const float s_ox = __shfl(ox, src, 64), s_oy = __shfl(oy, src, 64), s_oz = __shfl(oz, src, 64);
                const float s_ix = __shfl(ix, src, 64), s_iy = __shfl(iy, src, 64), s_iz = __shfl(iz, src, 64);
                const float s_dx = __shfl(d.x, src, 64), s_dy = __shfl(d.y, src, 64), s_dz = __shfl(d.z, src, 64);
                const float s_tmin = __shfl(t_min, src, 64), s_tmax = __shfl(t_max, src, 64);
                const uint32_t s_ridx = (uint32_t)__shfl((int)ridx, src, 64);
                const int s_base = __shfl(sbase, src, 64);
                if (take) {
                    ox = s_ox; oy = s_oy; oz = s_oz; ix = s_ix; iy = s_iy; iz = s_iz; d = V3(s_dx, s_dy, s_dz); ro = V3(s_ox, s_oy, s_oz);
                    rc = ray_margins(scene_bs, ox, oy, oz, ix, iy, iz);
                    t_min = s_tmin; t_max = s_tmax; ridx = s_ridx;
                    cur = (int)lds[s_base * MR_TRACE_BLOCK + ((threadIdx.x & ~63) | src)];
                    sp = 0; sbase = 0; have = true;
                }
                if (give) sbase++;
            }
        }
        if (need && !exhausted) {
            if (chunk_next >= chunk_end) exhausted = !grab_chunk(work_head, n, q_per, chunk, q_cur, q_fail, q_known, chunk_next, chunk_end, lane);
            if (!exhausted) {
                const uint32_t idx0 = chunk_next + (uint32_t)__popcll(need & lt_mask);
                const uint32_t idx = idx0;
                if (!have && idx0 < chunk_end) {
                    float4 a, b; bool dead = false;
                    if (SRC == 1) {
                        const uint2 it = reinterpret_cast<const uint2*>(rays)[idx >> 1];      // pair (a, b): even ray a -> b's light, odd ray b -> a's light
                        const uint32_t op = (idx & 1u) ? it.y : it.x, lp = (idx & 1u) ? it.x : it.y;
                        const float4 P = src.grec[4 * (size_t)op + 3], L = src.rrec[2 * (size_t)lp];
                        dead = src.skip_dead && L.w == 0.f;   // engine.hpp RaySrc: the light sample's carried luminance is 0 — nobody can see this ray's answer
#if MR_ANY_LEANREFILL
                        const v3 dir = oct_decode_lean(V2(L.y, L.z));
#else
                        const v3 dir = oct_decode(V2(L.y, L.z));
#endif
                        v3 o = V3(P.x, P.y, P.z) + src.vis_near * dir;          // put_ray (passes.hip): the same two expressions
                        a.x = o.x; a.y = o.y; a.z = o.z; a.w = 0.f; b.x = dir.x; b.y = dir.y; b.z = dir.z; b.w = 1e7f;
                    } else { a = reinterpret_cast<const float4*>(rays + idx)[0]; b = reinterpret_cast<const float4*>(rays + idx)[1]; }
                    ridx = idx; ro = V3(a.x, a.y, a.z); t_min = a.w; t_max = b.w;
#if MR_ANY_LEANREFILL
                    // Round 6: the pixel-pair source forms its own rays (an oct-decoded direction: components are 0 or at least 2^-25 in magnitude, the squared length lies in
                    // [1/3, 3]), so every operand of the normalisation and of the three reciprocals is inside the range in which the short sequences return the IEEE bits
                    // (device_math.hpp: proved by exhaustion, mirres_selfcheck_arith). Rays that come through the API keep the compiler's division / square root.
                    if (SRC == 1) d = normalize_lean(V3(b.x, b.y, b.z)); else d = normalize(V3(b.x, b.y, b.z));
#else
                    d = normalize(V3(b.x, b.y, b.z));
#endif
                    ox = ro.x; oy = ro.y; oz = ro.z;
                    { float dx = d.x, dy = d.y, dz = d.z;
                      if (dx == 0.f) dx = 0.000001f; if (dy == 0.f) dy = 0.000001f; if (dz == 0.f) dz = 0.000001f;
#if MR_ANY_LEANREFILL
                      if (SRC == 1) { ix = lean_rcp(dx); iy = lean_rcp(dy); iz = lean_rcp(dz); } else
#endif
                      { ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz; } }
                    sp = 0; sbase = 0;
                    rc = ray_margins(scene_bs, ox, oy, oz, ix, iy, iz);
                    hit_out[idx] = 0;          // set to 1 by whichever lane finds an occluder (the owner or, in the tail, a helper)
                    const float o3[3] = {ox, oy, oz}, i3[3] = {ix, iy, iz};
                    Slab s0 = slab(B.root_box, B.root_box + 3, o3, i3, t_min);
                    if (COUNT) c_boxes++;
                    if (!dead && s0.tf > s0.tn && t_max > s0.tn) { cur = TOPN > 0 ? MR_TOPBIT : 0; have = true; }
                }
                const uint32_t want = (uint32_t)__popcll(need);
                chunk_next = (chunk_next + want < chunk_end) ? chunk_next + want : chunk_end;
            }
        }
        if (!__ballot(have)) { if (exhausted) break; else continue; }
        do {
            int ref = cur; bool active = have;
            if (COUNT) { const uint64_t lm = __ballot(active && ref < 0); w_iters++; if (lm) { w_leaf_iters++; w_leaf_lanes += (unsigned long long)__popcll(lm); } }
            if (active) {
                // one 64-byte record per iteration — a Node4q or a LeafRec — fetched before the type is looked at, so that a wave pays ONE memory
                // round trip per iteration however its lanes split between nodes and leaves (the slowest lane sets the wave's pace)
                uint4 h0, h1, h2, rf;
                const bool leaf = ref < 0;
                if (TOPN > 0 && !leaf && (ref & MR_TOPBIT)) {
                    const uint4* nd = s_top + (size_t)(ref & 0xffff) * 4;
                    h0 = nd[0]; h1 = nd[1]; h2 = nd[2]; rf = nd[3];
                } else {
                    const uint4* __restrict__ nd = leaf ? reinterpret_cast<const uint4*>(B.leaves + ~ref) : reinterpret_cast<const uint4*>(B.nodes4q + ref);
                    h0 = nd[0]; h1 = nd[1]; h2 = nd[2]; rf = nd[3];
                    if (COUNT) c_nodes++;    // 64-byte records fetched from global memory (nodes served from LDS are not charged)
                }
                bool hit = false;
                int next = 0x7fffffff; float next_tn = 0.f;
                if (leaf) {
                    // the reference visits a leaf iff its OWN box passes: re-test with the exact box kept beside the triangle
                    const float4 l0 = make_float4(__uint_as_float(h0.x), __uint_as_float(h0.y), __uint_as_float(h0.z), __uint_as_float(h0.w));
                    const float4 l1 = make_float4(__uint_as_float(h1.x), __uint_as_float(h1.y), __uint_as_float(h1.z), __uint_as_float(h1.w));
                    const float4 l2 = make_float4(__uint_as_float(h2.x), __uint_as_float(h2.y), __uint_as_float(h2.z), __uint_as_float(h2.w));
                    const float4 l3 = make_float4(__uint_as_float(rf.x), __uint_as_float(rf.y), __uint_as_float(rf.z), __uint_as_float(rf.w));
                    const float ex0 = (l2.y - ox) * ix, ex1 = (l3.x - ox) * ix;
                    const float ey0 = (l2.z - oy) * iy, ey1 = (l3.y - oy) * iy;
                    const float ez0 = (l2.w - oz) * iz, ez1 = (l3.z - oz) * iz;
                    const bool nx_ = ix < 0.f, ny_ = iy < 0.f, nz_ = iz < 0.f;           // the reference's swap (see slab(): a NaN plane stays in its slot)
                    const float etn = fmaxf(fmaxf(fmaxf(nx_ ? ex1 : ex0, ny_ ? ey1 : ey0), nz_ ? ez1 : ez0), t_min);
                    const float etf = fminf(fminf(nx_ ? ex0 : ex1, ny_ ? ey0 : ey1), nz_ ? ez0 : ez1);
                    if (COUNT) c_boxes++;
                    if (etf > etn && t_max > etn) { hit = tri_accepts_regs<TIMED == 2>(l0, l1, l2, ro, d); if (COUNT) c_leaves++; }
                } else {
                    // conservative boxes: one fma per plane (node_cons / child_slab above)
                    const NodeCons nc = node_cons(h0, h1, h2, ox, oy, oz, ix, iy, iz, rc);
                    const int ref[4] = {(int)rf.x, (int)rf.y, (int)rf.z, (int)rf.w};
                    float tn4[4], tf4[4];
                    child_slabs(nc, t_min, tn4, tf4);
                    // A node defers at most three entries. While all three fit the LDS part of the stack (always, on trees of ordinary depth) the
                    // per-entry range checks of the general path — two compares and a select each, 4 issue cycles apiece — are not needed.
#if MR_ANY_SEL
                    // Round 6: child selection as straight-line code. The nearest passing child is the minimum of four keys — the entry distance's bit pattern
                    // as a signed integer (floats order like their bits where it matters: non-negative distances below +inf; a negative one, a caller's negative t_min, merely counts
                    // as nearest; no NaN canonicalisation in front of an integer minimum), +inf for a child that fails — and the highest index among equal minima; the up to three
                    // other passing children are stored UNCONDITIONALLY at the lane's stack top, which only advances past an entry that is to be kept (what lies above `sp` is
                    // never read; the LDS part has one spare row for the last store). The lane-mask logic is spelled out on the 64-bit masks themselves (ballot / inverse ballot:
                    // scalar and / andn2 — the compiler turns a negated compare into a second v_cmp otherwise). The deferred entries are in slot order instead of the sequential
                    // insert's "swap with the nearest so far"; a shadow ray's answer is an OR over subtrees, any order gives the same bit. Round 5's code: -DMR_ANY_SEL=0.
                    // Measured (profiles/r06_ab_any_sel*.txt): kernel -4.5 % / -10 % (icosphere / lego-like), frame +2.8 % / +3.5 % with the short refill arithmetic; the
                    // variants with FEWER VALU instructions — exec-masked stores (2), a compare tournament (3), v_addc on the mask (6) — were all slower than this one.
                    if (sp + 3 <= MR_ANY_LDS) {
                        const int INFI = 0x7f800000;
                        const uint64_t o0 = __builtin_amdgcn_ballot_w64(tf4[0] > tn4[0]), o1 = __builtin_amdgcn_ballot_w64(tf4[1] > tn4[1]);
                        const uint64_t o2 = __builtin_amdgcn_ballot_w64(tf4[2] > tn4[2]), o3 = __builtin_amdgcn_ballot_w64(tf4[3] > tn4[3]);
                        const int k0 = __builtin_amdgcn_inverse_ballot_w64(o0) ? __float_as_int(tn4[0]) : INFI, k1 = __builtin_amdgcn_inverse_ballot_w64(o1) ? __float_as_int(tn4[1]) : INFI;
                        const int k2 = __builtin_amdgcn_inverse_ballot_w64(o2) ? __float_as_int(tn4[2]) : INFI, k3 = __builtin_amdgcn_inverse_ballot_w64(o3) ? __float_as_int(tn4[3]) : INFI;
                        const int m = min(min(min(k0, k1), k2), k3);
                        const uint64_t e1 = __builtin_amdgcn_ballot_w64(k1 == m), e2 = __builtin_amdgcn_ballot_w64(k2 == m), e3 = __builtin_amdgcn_ballot_w64(k3 == m);
                        int nx = ref[0]; nx = __builtin_amdgcn_inverse_ballot_w64(e1) ? ref[1] : nx; nx = __builtin_amdgcn_inverse_ballot_w64(e2) ? ref[2] : nx; nx = __builtin_amdgcn_inverse_ballot_w64(e3) ? ref[3] : nx;
                        // the chosen child: the highest index among the minima; the others that pass are kept
                        const uint64_t p3 = o3 & ~e3, p2 = o2 & ~(e2 & ~e3), p1 = o1 & ~(e1 & ~(e2 | e3)), p0 = o0 & (e1 | e2 | e3);
                        if (COUNT) { for (int k = 0; k < 4; k++) if (ref[k] != ~B.T) c_boxes++; }
                        lds_stack[sp * MR_TRACE_BLOCK] = (uint32_t)ref[0]; sp += __builtin_amdgcn_inverse_ballot_w64(p0) ? 1 : 0;
                        lds_stack[sp * MR_TRACE_BLOCK] = (uint32_t)ref[1]; sp += __builtin_amdgcn_inverse_ballot_w64(p1) ? 1 : 0;
                        lds_stack[sp * MR_TRACE_BLOCK] = (uint32_t)ref[2]; sp += __builtin_amdgcn_inverse_ballot_w64(p2) ? 1 : 0;
                        lds_stack[sp * MR_TRACE_BLOCK] = (uint32_t)ref[3]; sp += __builtin_amdgcn_inverse_ballot_w64(p3) ? 1 : 0;
                        next = __builtin_amdgcn_inverse_ballot_w64(o0 | o1 | o2 | o3) ? nx : 0x7fffffff;
                    } else {
#else
                    if (sp + 3 <= MR_ANY_LDS) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float tn = tn4[k], tf = tf4[k];
                            const bool ok = tf > tn;
                            if (COUNT && ref[k] != ~B.T) c_boxes++;
                            if (ok) {
                                if (next == 0x7fffffff) { next = ref[k]; next_tn = tn; }
                                else {
                                    int far = ref[k];
                                    if (tn < next_tn) { far = next; next = ref[k]; next_tn = tn; }
                                    lds_stack[sp * MR_TRACE_BLOCK] = (uint32_t)far; sp++;
                                }
                            }
                        }
                    } else {
#endif
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float tn = tn4[k], tf = tf4[k];
                        // interior boxes only steer: `t_max > tn` (a pure cull; the leaf's exact test applies it) is left out — every compare is 4 issue
                        // cycles per wave (scripts/ubench/valu_rates.hip) and shadow rays towards the environment have no far limit to cull with.
                        // Unused slots (lo = 255 > hi = 0) fail the decoded test by themselves; with the margins a node thinner than 2^-19 of the
                        // scene on all three axes could let one through — it then leads to the null leaf (leaves[T]), whose exact box no ray passes.
                        const bool ok = tf > tn;
                        if (COUNT && ref[k] != ~B.T) c_boxes++;
                        if (ok) {
                            if (next == 0x7fffffff) { next = ref[k]; next_tn = tn; }
                            else {
                                int far = ref[k];
                                if (tn < next_tn) { far = next; next = ref[k]; next_tn = tn; }   // keep the nearest (node or leaf) for the immediate descent
                                if (sp < MR_ANY_LDS) lds_stack[sp * MR_TRACE_BLOCK] = (uint32_t)far;
                                else if (sp < MR_ANY_STACK) spill[sp - MR_ANY_LDS] = (uint32_t)far;
                                if (sp < MR_ANY_STACK) sp++;
                                else {    // cannot happen (bound above, static_assert); were it to, it is not silent: the sticky word makes mirres_render / mirres_bvh_trace fail, mirres_ctx_stats()[11] counts
                                    if (B.err) __hip_atomic_store(B.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    if (stats) atomicAdd(&stats[11], 1ull);
                                }
                            }
                        }
                    }
                    }
                }
                bool done = hit;
#if MR_ANY_POP
                // Round 6: the pop as straight-line code while the entry it would take lies in the LDS part of the stack (sp <= MR_ANY_LDS): the top entry is read
                // whether or not it is needed (one ds_read per iteration; what it returns for an empty stack is not used), the next reference, the stack pointer and
                // "done" are selects — no exec-mask region for "descend / pop / finished"
                if (sp <= MR_ANY_LDS) {
                    const bool has_next = next != 0x7fffffff, can_pop = sp > sbase;
                    const int top = (int)lds_stack[(sp > 0 ? sp - 1 : 0) * MR_TRACE_BLOCK];
                    cur = has_next ? next : top;
                    sp -= (!has_next && can_pop) ? 1 : 0;
                    done = hit || (!has_next && !can_pop);
                } else
#endif
                if (!hit) {
                    if (next != 0x7fffffff) cur = next;
                    else if (sp > sbase) { --sp; cur = (int)((sp < MR_ANY_LDS) ? lds_stack[sp * MR_TRACE_BLOCK] : spill[sp - MR_ANY_LDS]); }
                    else done = true;
                }
                if (COUNT) c_maxsp = sp > c_maxsp ? sp : c_maxsp;
                if (done) { have = false; if (hit) hit_out[ridx] = 1; }
            }
            // in the tail leave the loop as soon as an idle lane and a lane with deferred work coexist (hand-over above)
        } while (exhausted ? (__ballot(have) && !(__ballot(!have) && __ballot(have && sp > sbase && sbase < MR_ANY_LDS))) : (__popcll(__ballot(have)) >= MR_REFILL));
    }
    if (B.dbg && (threadIdx.x & 63) == 0) B.dbg[2 * (blockIdx.x * (MR_TRACE_BLOCK / 64) + (threadIdx.x >> 6)) + 1] = wall_clock64();
    if (COUNT && stats) { atomicAdd(&stats[2], c_boxes); atomicAdd(&stats[3], c_nodes); atomicAdd(&stats[4], c_leaves); atomicMax(&stats[8], (unsigned long long)c_maxsp);
                          if (lane == 0) { atomicAdd(&stats[13], w_iters); atomicAdd(&stats[14], w_leaf_iters); atomicAdd(&stats[15], w_leaf_lanes); } }
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[0], (unsigned long long)n);
}


// ---------------------------------------------------------------- closest hit: ordered 4-wide fast path + exact fallback
// The reference's closest-hit result depends on its (unordered, right-first) visiting order only in two situations:
//   (a) a triangle is accepted with t <= t_min (triangle_hit ignores the t interval, helperDi.slang:172-195): `closest` drops to <= t_min and the
//       rest of the search is culled, so WHICH behind-the-origin triangle is found first matters;
//   (b) two different triangles are accepted with exactly the same t (the later one in visiting order provides prim / normal).
// Case (a) is common on real meshes (46 % of the indirect rays of the lego-like scene: a large triangle's box contains the origins of the rays leaving the
// faces around it) and it has an order-free answer (round 4). bvh_hit pushes left then right and pops the right child first, and leaf slots are the
// sorted Morton positions, so the reference reaches leaves in DESCENDING slot order. A triangle accepted behind the origin has the origin inside its
// leaf box along the line: its entry distance is t_min itself, smaller than any `closest` that positive hits can have produced, so it is never culled —
// neither by the reference before it gets there nor by this kernel. Hence: if any triangle is accepted with t <= t_min, the reference's search ends at the
// one with the HIGHEST slot, and reports that triangle (t = closest = its t, prim, normal: `now_t <= closest`). This kernel keeps the highest such slot
// (`neg_t` set), does not let those hits shrink `closest`, and from the first of them on only looks into boxes that contain the origin. The rare leaf whose
// computed entry distance is not exactly t_min although its triangle's t is <= t_min (rounding of two different expressions) still goes to the redo list.
// Otherwise every traversal that culls with `closest > tn` finds the same minimum: the minimum triangle's boxes all contain its hit point,
// so their entry distance tn <= t_min_hit <= closest and they are never culled (DESIGN.md §Traversal exactness). This kernel therefore walks
// the 4-wide collapse front to back (far more culling than the reference order), watches for (a) and (b), and appends the few affected rays
// to a redo list that k_trace_persist<false> (reference order) then recomputes. The final output is bit-identical to the reference order.
template <bool COUNT>
__global__ void __launch_bounds__(MR_TRACE_BLOCK) k_trace_closest4(BvhView B, const Ray* __restrict__ rays, const uint32_t* __restrict__ d_count,
                                                                   uint32_t n_fixed, uint32_t* __restrict__ work_head, HitRec* __restrict__ rec,
                                                                   int32_t* __restrict__ hit_out, float* __restrict__ t_out, float* __restrict__ pos_out,
                                                                   float* __restrict__ nrm_out, int32_t* __restrict__ prim_out,
                                                                   uint32_t* __restrict__ redo, uint32_t* __restrict__ redo_count,
                                                                   unsigned long long* __restrict__ stats) {
    // Same machinery as k_trace_any4q: compressed 4-wide nodes (conservative boxes only steer and cull: a culled subtree's entry distance is
    // <= the exact one, so nothing that could beat `closest` is skipped), leaves re-tested against their exact box, ONE 64-byte record per
    // lane and iteration fetched before its type is known, results written after the loop. Children are visited front to back and the stack
    // keeps (reference, entry distance) so that deferred subtrees are re-tested against the then-current `closest`.
    __shared__ uint2 lds[MR_LDS_STACK * MR_TRACE_BLOCK];
    uint2* const lds_stack = lds + threadIdx.x;
    const uint32_t n = d_count ? *d_count : n_fixed;
    const int lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t chunk = n / (gridDim.x * (MR_TRACE_BLOCK / 64) * MR_CHUNK_DIV);
    chunk = chunk < 64 ? 64 : (chunk > MR_CHUNK_MAX ? MR_CHUNK_MAX : (chunk & ~63u));
    const uint32_t q_per = (((n + MR_NQ - 1) / MR_NQ) + 63u) & ~63u;                                  // rays per sub-queue
    uint32_t q_cur = (blockIdx.x * (MR_TRACE_BLOCK / 64) + (threadIdx.x >> 6)) % MR_NQ, q_fail = 0;   // wave-uniform
    uint32_t q_known = 0;   // sub-queues this wave knows to be empty (grab_chunk)
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false, have = false, fin = false;
    float ox = 0, oy = 0, oz = 0, ix = 0, iy = 0, iz = 0; v3 d = V3(0.f), ro = V3(0.f);
    float t_min = 0.f, closest = 0.f, best_u = 0.f, best_v = 0.f;
    RayCons rc; rc.mx = rc.my = rc.mz = 0.f;
    const float scene_bs = scene_bound(B.root_box);
    const int NONE = 0x7fffffff;
    int cur = NONE, sp = 0, best_slot = -1; uint32_t ridx = 0; bool any_hit = false, need_redo = false;
    const float NO_NEG = 3.0e38f;
    float neg_t = NO_NEG;      // case (a): t of the accepted triangle with t <= t_min and the highest slot (then best_slot / best_u / best_v are that triangle's); NO_NEG: none yet
    uint2 spill[MR_STACK - MR_LDS_STACK];
    unsigned long long c_boxes = 0, c_nodes = 0, c_leaves = 0; int c_maxsp = 0;
    while (true) {
        const uint64_t need = __ballot(!have);
        if (need && !exhausted) {
            if (chunk_next >= chunk_end) exhausted = !grab_chunk(work_head, n, q_per, chunk, q_cur, q_fail, q_known, chunk_next, chunk_end, lane);
            if (!exhausted) {
                const uint32_t idx = chunk_next + (uint32_t)__popcll(need & lt_mask);
                if (!have && idx < chunk_end) {
                    const float4 a = reinterpret_cast<const float4*>(rays + idx)[0], b = reinterpret_cast<const float4*>(rays + idx)[1];
                    ridx = idx; ro = V3(a.x, a.y, a.z); t_min = a.w; closest = b.w;
                    d = normalize(V3(b.x, b.y, b.z));
                    ox = ro.x; oy = ro.y; oz = ro.z;
                    { float dx = d.x, dy = d.y, dz = d.z;
                      if (dx == 0.f) dx = 0.000001f; if (dy == 0.f) dy = 0.000001f; if (dz == 0.f) dz = 0.000001f;
                      ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz; }
                    sp = 0; any_hit = false; need_redo = false; best_u = 0.f; best_v = 0.f; best_slot = -1; neg_t = NO_NEG;
                    rc = ray_margins(scene_bs, ox, oy, oz, ix, iy, iz);
                    const float o3[3] = {ox, oy, oz}, i3[3] = {ix, iy, iz};
                    Slab s0 = slab(B.root_box, B.root_box + 3, o3, i3, t_min);
                    if (COUNT) c_boxes++;
                    cur = (s0.tf > s0.tn && closest > s0.tn) ? 0 : NONE;
                    have = true;
                }
                const uint32_t want = (uint32_t)__popcll(need);
                chunk_next = (chunk_next + want < chunk_end) ? chunk_next + want : chunk_end;
            }
        }
        if (!__ballot(have)) { if (exhausted) break; else continue; }
        do {
            if (have) {
                bool done = false, skip = false;
                if (cur == NONE) {   // pop the nearest deferred subtree that still beats `closest`
                    bool found = false;
                    // at most one pop per iteration: a culled entry costs this lane an iteration instead of making the whole wave wait for the longest run of
                    // culled entries (rounds 1-4 popped until an entry survived)
                    if (sp > 0) {
                        --sp;
                        const uint2 e = (sp < MR_LDS_STACK) ? lds_stack[sp * MR_TRACE_BLOCK] : spill[sp - MR_LDS_STACK];
                        const float etn = __uint_as_float(e.y);
                        if (closest > etn) { cur = (int)e.x; found = true; }
                        else { if (closest == etn && any_hit) need_redo = true; found = true; skip = true; }
                    }
                    if (!found) done = true;
                }
                if (!done && !skip) {
                    const bool leaf = cur < 0;
                    const uint4* __restrict__ nd = leaf ? reinterpret_cast<const uint4*>(B.leaves + ~cur) : reinterpret_cast<const uint4*>(B.nodes4q + cur);
                    const uint4 h0 = nd[0], h1 = nd[1], h2 = nd[2], rf = nd[3];
                    if (COUNT) c_nodes++;
                    if (leaf) {
                        const int slot = ~cur;
                        cur = NONE;
                        const float4 l0 = make_float4(__uint_as_float(h0.x), __uint_as_float(h0.y), __uint_as_float(h0.z), __uint_as_float(h0.w));
                        const float4 l1 = make_float4(__uint_as_float(h1.x), __uint_as_float(h1.y), __uint_as_float(h1.z), __uint_as_float(h1.w));
                        const float4 l2 = make_float4(__uint_as_float(h2.x), __uint_as_float(h2.y), __uint_as_float(h2.z), __uint_as_float(h2.w));
                        const float4 l3 = make_float4(__uint_as_float(rf.x), __uint_as_float(rf.y), __uint_as_float(rf.z), __uint_as_float(rf.w));
                        // the reference tests a leaf iff its OWN box passes against the current `closest`
                        const float ex0 = (l2.y - ox) * ix, ex1 = (l3.x - ox) * ix;
                        const float ey0 = (l2.z - oy) * iy, ey1 = (l3.y - oy) * iy;
                        const float ez0 = (l2.w - oz) * iz, ez1 = (l3.z - oz) * iz;
                        const bool nx_ = ix < 0.f, ny_ = iy < 0.f, nz_ = iz < 0.f;       // the reference's swap (see slab(): a NaN plane stays in its slot)
                        const float etn = fmaxf(fmaxf(fmaxf(nx_ ? ex1 : ex0, ny_ ? ey1 : ey0), nz_ ? ez1 : ez0), t_min);
                        const float etf = fminf(fminf(nx_ ? ex0 : ex1, ny_ ? ey0 : ey1), nz_ ? ez0 : ez1);
                        if (COUNT) c_boxes++;
                        if (etf > etn && closest == etn && any_hit) need_redo = true;
                        if (etf > etn && closest > etn) {
                            if (COUNT) c_leaves++;
                            const v3 v0 = V3(l0.x, l0.y, l0.z), E1 = V3(l0.w, l1.x, l1.y), E2 = V3(l1.z, l1.w, l2.x);
                            const v3 P = cross(d, E2);
                            const float det = dot(E1, P);
                            if (!(det > -1e-15f && det < 1e-15f)) {
                                const float invDet = 1 / det;
                                const v3 Tv = ro - v0;
                                const float u = dot(Tv, P) * invDet;
                                if (!(u < 0 || u > 1)) {
                                    const v3 Q = cross(Tv, E1);
                                    const float v = dot(d, Q) * invDet;
                                    if (!(v < 0 || u + v > 1)) {
                                        const float t = dot(E2, Q) * invDet;
                                        any_hit = true;
                                        if (t <= t_min) {                                                               // case (a): see the head comment
                                            if (etn != t_min) need_redo = true;                                         // entry distance and t disagree about the origin: let the reference order decide
                                            if (neg_t == NO_NEG || slot > best_slot) { best_slot = slot; best_u = u; best_v = v; neg_t = t; }
                                            // nothing in front of the origin matters any more: only boxes that contain the origin (entry distance t_min) can hold
                                            // another such triangle. (`closest` is no longer a hit distance and positive hits are no longer recorded.)
                                            closest = fminf(closest, fmaxf(t_min, 0.f) + 1.17549435e-38f);
                                        } else if (!(t > 0.f)) need_redo = true;                                        // NaN (or t_min < t <= 0 for a caller's negative t_min)
                                        else if (neg_t == NO_NEG && t <= closest) {
                                            if (t == closest && best_slot >= 0 && best_slot != slot) need_redo = true;    // case (b)
                                            closest = t; best_u = u; best_v = v; best_slot = slot;
                                        }
                                    }
                                }
                            }
                        }
                    } else {
                        // conservative boxes: one fma per plane (node_cons / child_slab); a smaller entry distance culls less and never wrongly
                        const NodeCons nc = node_cons(h0, h1, h2, ox, oy, oz, ix, iy, iz, rc);
                        const int ref[4] = {(int)rf.x, (int)rf.y, (int)rf.z, (int)rf.w};
                        float tn4[4], tf4[4];
                        child_slabs(nc, t_min, tn4, tf4);
                        // children that may still hold something nearer, nearest first: a five-comparator sorting network over (entry distance, reference), a child that
                        // cannot matter carrying +inf (round 5; rounds 1-4 inserted one child after the other into a sorted list: sixteen predicated swaps of which the
                        // compiler keeps all — the node branch of this kernel was nearly twice the shadow-ray kernel's). Any order gives the same minimum (see above).
                        float sk[4]; int sr[4]; int nn = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool ok = tf4[k] > tn4[k] && closest > tn4[k];   // unused slots: see k_trace_any4q
                            if (COUNT && ref[k] != ~B.T) c_boxes++;
                            sk[k] = ok ? tn4[k] : __builtin_inff(); sr[k] = ref[k]; nn += ok ? 1 : 0;
                        }
#define MR_CX(a, b) { const bool sw_ = sk[b] < sk[a]; const float ka_ = sw_ ? sk[b] : sk[a], kb_ = sw_ ? sk[a] : sk[b]; const int ra_ = sw_ ? sr[b] : sr[a], rb_ = sw_ ? sr[a] : sr[b]; sk[a] = ka_; sk[b] = kb_; sr[a] = ra_; sr[b] = rb_; }
                        MR_CX(0, 1) MR_CX(2, 3) MR_CX(0, 2) MR_CX(1, 3) MR_CX(1, 2)
#undef MR_CX
                        cur = NONE;
#if MR_CL_SEL
                        // Round 6 (as in the shadow-ray kernel): while three entries fit the LDS part the stores are UNCONDITIONAL — farthest first, at a stack top that only
                        // advances past an entry to be kept. The sorted keys put the children that cannot matter (+inf) last, i.e. FIRST in push order: each is overwritten
                        // by the next store, and what lies above `sp` is never read. No exec-mask region per entry; the order of the kept entries is the old one.
                        if (sp + 3 <= MR_LDS_STACK) {
                            cur = nn > 0 ? sr[0] : NONE;
#pragma unroll
                            for (int q = 3; q >= 1; q--) {
                                uint2 e; e.x = (uint32_t)sr[q]; e.y = __float_as_uint(sk[q]);
                                lds_stack[sp * MR_TRACE_BLOCK] = e; sp += (q < nn) ? 1 : 0;
                            }
                            if (COUNT) c_maxsp = sp > c_maxsp ? sp : c_maxsp;
                        } else
#endif
                        if (nn > 0) {
                            cur = sr[0];
                            if (sp + 3 <= MR_LDS_STACK) {      // all three possible entries fit the LDS part (the usual case): no per-entry range checks
#pragma unroll
                                for (int q = 3; q >= 1; q--) {
                                    if (q < nn) { uint2 e; e.x = (uint32_t)sr[q]; e.y = __float_as_uint(sk[q]); lds_stack[sp * MR_TRACE_BLOCK] = e; sp++; }
                                }
                            } else {
#pragma unroll
                            for (int q = 3; q >= 1; q--) {      // farthest first: the nearest deferred child is popped first
                                if (q < nn) {
                                    uint2 e; e.x = (uint32_t)sr[q]; e.y = __float_as_uint(sk[q]);
                                    if (sp < MR_LDS_STACK) lds_stack[sp * MR_TRACE_BLOCK] = e;
                                    else if (sp < MR_STACK) spill[sp - MR_LDS_STACK] = e;
                                    if (sp < MR_STACK) sp++; else need_redo = true;   // a full stack hands the ray to the reference-order kernel, whose stack cannot overflow
                                }
                            }
                            }
                            if (COUNT) c_maxsp = sp > c_maxsp ? sp : c_maxsp;
                        }
                    }
                }
                if (done) { have = false; fin = true; }
            }
        } while (__popcll(__ballot(have)) >= MR_CL_REFILL || (exhausted && __ballot(have)));
        if (fin) {   // results of the rays that finished in this stretch, written by all of them together
            fin = false;
            if (need_redo) redo[atomicAdd(redo_count, 1u)] = ridx;
            TraceOut r; r.hit = any_hit; r.t = any_hit ? closest : 0.f; r.u = best_u; r.v = best_v; r.slot = best_slot; r.d = d;
            if (neg_t != NO_NEG) r.t = neg_t;      // case (a): the reference's search ended at that triangle
            v3 p, nn_; int pr;
            finish_closest(B, r, ro, p, nn_, pr);
            if (rec) {
                float4 o0, o1;
                o0.x = p.x; o0.y = p.y; o0.z = p.z; o0.w = __int_as_float(any_hit ? 1 : 0);
                o1.x = nn_.x; o1.y = nn_.y; o1.z = nn_.z; o1.w = r.t;
                reinterpret_cast<float4*>(rec + ridx)[0] = o0; reinterpret_cast<float4*>(rec + ridx)[1] = o1;
            }
            if (hit_out) hit_out[ridx] = any_hit ? 1 : 0;
            if (t_out) t_out[ridx] = r.t;
            if (pos_out) st3(pos_out, ridx, p);
            if (nrm_out) st3(nrm_out, ridx, nn_);
            if (prim_out) prim_out[ridx] = pr;
        }
    }
    if (COUNT && stats) { atomicAdd(&stats[5], c_boxes); atomicAdd(&stats[6], c_nodes); atomicAdd(&stats[7], c_leaves); atomicMax(&stats[9], (unsigned long long)c_maxsp); }
    if (stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[1], (unsigned long long)n);
}

static int persist_grid(size_t capacity);
static int ensure_redo(mirres_bvh* bvh, int alt, size_t capacity) {
    if (bvh->redo_cap[alt] >= capacity) return 0;
    if (bvh->redo[alt]) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipFree(bvh->redo[alt])); bvh->redo[alt] = nullptr; }
    MR_HIP(hipMalloc(&bvh->redo[alt], sizeof(uint32_t) * capacity));
    bvh->redo_cap[alt] = capacity;
    return 0;
}
// ordered fast path + reference-order recomputation of the handed-back rays (see k_trace_closest4). `lane` 4 (the second path-tracing stream of
// mirres_render) works on its own head sets and redo list, so two such traces may be in flight
template <bool COUNT>
static int closest_fast(mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, HitRec* rec, int32_t* hit, float* t, float* pos,
                        float* nrm, int32_t* prim, unsigned long long* stats, hipStream_t s, int lane = 0) {
    const int alt = lane == 4 ? 1 : 0;
    int rc = ensure_redo(bvh, alt, capacity); if (rc) return rc;
    uint32_t* const w = bvh->work + (alt ? 12 : 4) * MR_WSET;   // set +0 fast heads, set +1 [0] redo count, set +2 redo heads
    MR_HIP(hipMemsetAsync(w, 0, 3 * MR_WSET * sizeof(uint32_t), s));
    k_trace_closest4<COUNT><<<persist_grid(capacity), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, w, rec, hit, t, pos, nrm, prim,
                                                                               bvh->redo[alt], w + MR_WSET, stats);
    k_trace_persist<false><<<256 * (256 / MR_TRACE_BLOCK), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, w + MR_WSET, 0u, w + 2 * MR_WSET, hit, rec, t, pos, nrm, prim, COUNT ? stats : nullptr, bvh->redo[alt]);
    MR_LAUNCH_CHECK("closest_fast");
    return 0;
}

static int g_timed_tag = 0;   // set by trace_any_queue for event-timed launches (mirres_ctx_set_instrument bit 1)
static int any_top();
template <bool COUNT>
static void launch_any4q(const mirres_bvh* bvh, int grid, const Ray* rays, const uint32_t* d_count, uint32_t cap, uint32_t* head, int32_t* hit,
                         unsigned long long* stats, hipStream_t s) {
    const int top = (bvh->T - 1 >= 341 * 4) ? any_top() : 0;
    if (top == 85 && !COUNT && g_timed_tag) k_trace_any4q<false, 85, 1><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, cap, head, hit, stats);
    else if (top == 85) k_trace_any4q<COUNT, 85><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, cap, head, hit, stats);
    else if (top == 341) k_trace_any4q<COUNT, 341><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, cap, head, hit, stats);
    else if (!COUNT && g_timed_tag) k_trace_any4q<false, 0, 1><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, cap, head, hit, stats);
    else k_trace_any4q<COUNT, 0><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, cap, head, hit, stats);
}
static int closest_mode() {   // 4 (default): ordered compressed 4-wide fast path + reference-order redo of the flagged rays; MIRRES_CLOSEST=2: reference order for all
    static int m = -1;
    if (m < 0) { const char* e = getenv("MIRRES_CLOSEST"); m = (e && e[0] == '2') ? 2 : 4; }
    return m;
}
static int persist_grid(size_t capacity) {
    size_t want = (capacity + MR_TRACE_BLOCK - 1) / MR_TRACE_BLOCK;
    static const size_t cap_blocks = [] { const char* e = getenv("MIRRES_TRACE_BLOCKS_PER_CU"); const int v = e ? atoi(e) : 8; return (size_t)(v >= 1 && v <= 16 ? v : 8); }();
    size_t cap = 256 * cap_blocks * (256 / MR_TRACE_BLOCK);      // workgroups (of 256 threads) launched per CU. Six are resident at once (75-79 VGPRs; forcing <= 64 spills, and the kernels are VALU-issue bound while CUs
                                        // hold waves, so more occupancy buys nothing); launching eight lets a CU that finishes early pick up another workgroup's share of the
                                        // queues' tails: +0.7 % (icosphere) / +0.9 % (lego-like) on the 512-spp frame against six, twelve and more lose on the icosphere
                                        // (profiles/r04_ab_trace_blocks.txt; rounds 1-4 launched six)
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

static int trace_grid(size_t capacity) {
    // persistent-style launch: enough workgroups (counted in 256 threads, whatever MR_TRACE_BLOCK is) to fill 256 CUs several times over, grid-stride beyond
    size_t want = (capacity + MR_TRACE_BLOCK - 1) / MR_TRACE_BLOCK;
    size_t cap = 256 * 8 * (256 / MR_TRACE_BLOCK);
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

// How many of the compressed tree's top nodes the shadow-ray kernel reads from LDS: 0 (default since the end of round 4), 85 (levels 0..3; the default of rounds
// 1-4) or 341 (MIRRES_TOPQ). Round 1 measured the 85-node prefix as a gain on the collapsed reference LBVH; on the final kernel and the private hierarchy it is a
// loss — launch -5 %, frame +3.5 % (icosphere) / +2.6 % (lego-like) without it (profiles/r04_ab_any_top.txt): two fetch paths in the hot loop cost more than the
// L1 hits they replace. The ordered closest-hit kernel never had it (and loses 3-5 % with it: r04_ab_closest_top.txt).
static int any_top() {
    static const int topq = [] { const char* e = getenv("MIRRES_TOPQ"); const int v = e ? atoi(e) : 0; return (v == 85 || v == 341) ? v : 0; }();
    return topq;
}
// the spatial pass's queue of (origin pixel, light pixel) pairs: same kernel, rays formed at the refill (head set of lane 0 ... 4 as below)
int trace_any_items_queue(const mirres_bvh* bvh, const uint2* items, const RaySrc& src, const uint32_t* d_count, size_t capacity, int32_t* hit,
                          unsigned long long* stats, hipStream_t s, int lane, int timed, bool heads_clean, int head_set) {
    static const int set_of_lane[5] = {0, 7, 9, 11, 15};
    uint32_t* const heads = bvh->work + (head_set >= 0 ? head_set : set_of_lane[lane]) * MR_WSET;      // head_set: a unit of the band pipeline (sets 0, 17, 18)
    if (!heads_clean) MR_HIP(hipMemsetAsync(heads, 0, MR_WSET * sizeof(uint32_t), s));
    const Ray* q = reinterpret_cast<const Ray*>(items);
    const int grid = persist_grid(capacity); const uint32_t cap = (uint32_t)capacity;
    const int top = (bvh->T - 1 >= 341 * 4 && any_top() == 85) ? 85 : 0;
    if (top == 85 && timed) k_trace_any4q<false, 85, 1, 1><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), q, d_count, cap, heads, hit, stats, src);
    else if (top == 85) k_trace_any4q<false, 85, 0, 1><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), q, d_count, cap, heads, hit, stats, src);
    else if (timed) k_trace_any4q<false, 0, 1, 1><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), q, d_count, cap, heads, hit, stats, src);
    else k_trace_any4q<false, 0, 0, 1><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), q, d_count, cap, heads, hit, stats, src);
    MR_LAUNCH_CHECK("trace_any_items_queue");
    return 0;
}
int trace_any_queue(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit,
                    unsigned long long* stats, hipStream_t s, int lane, int timed, bool heads_clean) {
    g_timed_tag = timed;
    static const int set_of_lane[5] = {0, 7, 9, 11, 15};                    // launches that may overlap on different streams use different head sets
    uint32_t* const heads = bvh->work + set_of_lane[lane] * MR_WSET;
    if (!heads_clean) MR_HIP(hipMemsetAsync(heads, 0, MR_WSET * sizeof(uint32_t), s));   // heads_clean: the previous consumer zeroed them (k_spatial_resolve in mirres_render's chain)
    launch_any4q<false>(bvh, persist_grid(capacity), rays, d_count, (uint32_t)capacity, heads, hit, stats, s);
    MR_LAUNCH_CHECK("trace_any_queue");
    return 0;
}
// occlusion with hits in front of the origin only (API head set 2): render_dump.py's batch_intersector
int trace_any_front_queue(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit, hipStream_t s) {
    uint32_t* const heads = bvh->work + 2 * MR_WSET;
    MR_HIP(hipMemsetAsync(heads, 0, MR_WSET * sizeof(uint32_t), s));
    const int grid = persist_grid(capacity);
    if (bvh->T - 1 >= 341 * 4 && any_top() == 85) k_trace_any4q<false, 85, 2><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, heads, hit, nullptr);
    else k_trace_any4q<false, 0, 2><<<grid, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, heads, hit, nullptr);
    MR_LAUNCH_CHECK("trace_any_front_queue");
    return 0;
}
int trace_closest_queue(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, HitRec* out,
                        unsigned long long* stats, hipStream_t s, int lane) {
    if (closest_mode() == 4) return closest_fast<false>(const_cast<mirres_bvh*>(bvh), rays, d_count, capacity, out, nullptr, nullptr, nullptr, nullptr, nullptr, stats, s, lane);
    static const int set_of_lane[5] = {1, 8, 10, 10, 16};
    uint32_t* const heads = bvh->work + set_of_lane[lane] * MR_WSET;
    MR_HIP(hipMemsetAsync(heads, 0, MR_WSET * sizeof(uint32_t), s));
    k_trace_persist<false><<<persist_grid(capacity), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, heads, nullptr, out,
                                                                             nullptr, nullptr, nullptr, nullptr, stats);
    MR_LAUNCH_CHECK("trace_closest_queue");
    return 0;
}
int trace_any_queue_counted(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit,
                            unsigned long long* stats, hipStream_t s, int reference_order) {
    if (reference_order) {   // visit counts of the reference's own traversal (bvh_hit order, no early exit) on the same rays — SURVEY §8d's accounting basis
        k_trace_any<true><<<trace_grid(capacity), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, hit, nullptr, stats);
        MR_LAUNCH_CHECK("trace_any_queue_counted(ref)");
        return 0;
    }
    MR_HIP(hipMemsetAsync(bvh->work, 0, MR_WSET * sizeof(uint32_t), s));
    launch_any4q<true>(bvh, persist_grid(capacity), rays, d_count, (uint32_t)capacity, bvh->work, hit, stats, s);
    MR_LAUNCH_CHECK("trace_any_queue_counted");
    return 0;
}
int trace_closest_queue_counted(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, HitRec* out,
                                unsigned long long* stats, hipStream_t s) {
    if (closest_mode() == 4) return closest_fast<true>(const_cast<mirres_bvh*>(bvh), rays, d_count, capacity, out, nullptr, nullptr, nullptr, nullptr, nullptr, stats, s);
    k_trace_closest<true><<<trace_grid(capacity), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), rays, d_count, (uint32_t)capacity, out, nullptr, nullptr,
                                                                           nullptr, nullptr, nullptr, nullptr, stats);
    MR_LAUNCH_CHECK("trace_closest_queue_counted");
    return 0;
}

}  // namespace mr

using namespace mr;

extern "C" int mirres_bvh_trace(mirres_bvh_t* bvh, const float* rays, int n, int mode, int32_t* hit, float* t, float* pos, float* normal,
                                int32_t* prim, uint32_t* counters, void* stream) {
    if (!bvh || !rays || n < 0 || mode < 0 || mode > 4) { set_error("mirres_bvh_trace: bad argument"); return MIRRES_E_ARG; }
    if (bvh->T < 2) { set_error("mirres_bvh_trace: BVH not built"); return MIRRES_E_STATE; }
    if (int e = bvh_sticky_error(bvh, "mirres_bvh_trace")) return e;
    if (n == 0) return MIRRES_OK;
    hipStream_t s = (hipStream_t)stream;
    const Ray* r = reinterpret_cast<const Ray*>(rays);
    const int g = trace_grid((size_t)n);
    if (mode == 2) return closest_fast<false>(bvh, r, nullptr, (size_t)n, nullptr, hit, t, pos, normal, prim, nullptr, s);
    if (mode == 3) {
        if (!hit) { set_error("mirres_bvh_trace: occlusion needs hit[]"); return MIRRES_E_ARG; }
        return trace_any_front_queue(bvh, r, nullptr, (size_t)n, hit, s);
    }
    if (mode == 4) {
        if (counters) { set_error("mirres_bvh_trace: mode 4 has no per-ray counters"); return MIRRES_E_ARG; }
        MR_HIP(hipMemsetAsync(bvh->work + 3 * MR_WSET, 0, MR_WSET * sizeof(uint32_t), s));
        k_trace_persist<false, true><<<persist_grid((size_t)n), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), r, nullptr, (uint32_t)n, bvh->work + 3 * MR_WSET, hit, nullptr, t, pos, normal,
                                                                                       prim, nullptr);
        MR_LAUNCH_CHECK("mirres_bvh_trace");
        return MIRRES_OK;
    }
    if (mode == 0) {
        if (!hit) { set_error("mirres_bvh_trace: any-hit needs hit[]"); return MIRRES_E_ARG; }
        if (counters) k_trace_any<true><<<g, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), r, nullptr, (uint32_t)n, hit, counters, nullptr);
        else {
            MR_HIP(hipMemsetAsync(bvh->work + 2 * MR_WSET, 0, MR_WSET * sizeof(uint32_t), s));
            launch_any4q<false>(bvh, persist_grid((size_t)n), r, nullptr, (uint32_t)n, bvh->work + 2 * MR_WSET, hit, nullptr, s);
        }
    } else {
        if (counters) k_trace_closest<true><<<g, MR_TRACE_BLOCK, 0, s>>>(bvh->view(), r, nullptr, (uint32_t)n, nullptr, hit, t, pos, normal, prim, counters, nullptr);
        else {
            MR_HIP(hipMemsetAsync(bvh->work + 3 * MR_WSET, 0, MR_WSET * sizeof(uint32_t), s));
            k_trace_persist<false><<<persist_grid((size_t)n), MR_TRACE_BLOCK, 0, s>>>(bvh->view(), r, nullptr, (uint32_t)n, bvh->work + 3 * MR_WSET, hit, nullptr, t, pos, normal,
                                                                                    prim, nullptr);
        }
    }
    MR_LAUNCH_CHECK("mirres_bvh_trace");
    return MIRRES_OK;
}

// development aid (not part of include/mirres.h): the PRODUCTION shadow-ray kernel in its counting build on n rays -> hit[n] and h_stats u64[12] in the layout of
// mirres_ctx_stats ([0] rays, [2..4] box tests / 64-byte records fetched / leaves tested, [8] deepest private stack, [11] stack overflows). Synchronises.
extern "C" int mirres_debug_any_stats(mirres_bvh_t* bvh, const float* rays, int n, int32_t* hit, unsigned long long* h_stats, void* stream) {
    if (!bvh || !rays || !hit || !h_stats || n <= 0 || bvh->T < 2) return MIRRES_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* d = nullptr;
    MR_HIP(hipMalloc(&d, 12 * sizeof(unsigned long long)));
    MR_HIP(hipMemsetAsync(d, 0, 12 * sizeof(unsigned long long), s));
    MR_HIP(hipMemsetAsync(bvh->work + 2 * MR_WSET, 0, MR_WSET * sizeof(uint32_t), s));
    launch_any4q<true>(bvh, persist_grid((size_t)n), reinterpret_cast<const Ray*>(rays), nullptr, (uint32_t)n, bvh->work + 2 * MR_WSET, hit, d, s);
    MR_HIP(hipStreamSynchronize(s));
    MR_HIP(hipMemcpy(h_stats, d, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    MR_HIP(hipFree(d));
    return MIRRES_OK;
}

// development aid (not part of include/mirres.h): enable != 0 allocates the per-wave timestamp buffer read by k_trace_any4q; out (host, 2*16384 u64) receives it
extern "C" int mirres_debug_wave_times(mirres_bvh_t* bvh, unsigned long long* out, int enable) {
    if (!bvh) return MIRRES_E_ARG;
    if (enable && !bvh->dbg) { MR_HIP(hipMalloc(&bvh->dbg, sizeof(unsigned long long) * 2 * 16384)); MR_HIP(hipMemset(bvh->dbg, 0, sizeof(unsigned long long) * 2 * 16384)); }
    if (out && bvh->dbg) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipMemcpy(out, bvh->dbg, sizeof(unsigned long long) * 2 * 16384, hipMemcpyDeviceToHost)); }
    if (!enable && bvh->dbg) { MR_HIP(hipFree(bvh->dbg)); bvh->dbg = nullptr; }
    return MIRRES_OK;
}
