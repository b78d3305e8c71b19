// bvh_build.hip — Karras-2012 LBVH build on gfx950.
// Behaviour contract: node arrays identical to the reference's 7-kernel build (nerf/bvhworkers/*.slang driven by
// restirbvhWorker.update_bvh, nerf/renderer_restir.py:25-89). MI355X redesign:
//   * scene extent by wave reduction + order-preserving atomics instead of 6 host-synchronising torch reductions;
//   * device-wide stable radix sort (k_rs_*: multi-workgroup, wave64 digit matching) instead of the single-256-thread-block sort
//     (lbvh_single_radixsort.slang, hard-coded 32-wide subgroups) — same stable ascending order;
//   * one-launch bottom-up refit with agent-scope arrival counters instead of tree_height launches + a host sync
//     (fmin/fmax unions are exact and order independent, so the boxes are bit-identical);
//   * a packing pass that emits the 64-byte two-child traversal records used by bvh_trace.hip.
#include "engine.hpp"
#include "device_math.hpp"
#include <cstring>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace mr {

static thread_local char g_err[512] = "";
const char* set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return g_err;
}
int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return MIRRES_E_HIP;
}

MR_DEV uint32_t f2ord(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
MR_DEV float ord2f(uint32_t u) { uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; return __uint_as_float(b); }

MR_DEV float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
MR_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// generateElements (get_elements.slang:3-39) + extent (renderer_restir.py:34-40). The six extent words share a cache line, so their atomics
// serialise (~88 per microsecond): one set per BLOCK of a 256-block grid-stride launch instead of one per wave (5 250 sets = 0.36 ms for 336 k triangles).
__global__ void __launch_bounds__(256) k_elements(const float* __restrict__ vert, const int32_t* __restrict__ tri, int T,
                                                  float* __restrict__ ele, uint32_t* __restrict__ extent) {
    __shared__ float s_mn[4][3], s_mx[4][3];
    float gmn[3] = {INFINITY, INFINITY, INFINITY}, gmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < T; p += gridDim.x * blockDim.x) {
        float mn[3] = {1e9f, 1e9f, 1e9f}, mx[3] = {-1e9f, -1e9f, -1e9f};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int vi = tri[3 * (size_t)p + i];
#pragma unroll
            for (int k = 0; k < 3; k++) { float v = vert[3 * (size_t)vi + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float a = fminf(mn[k], mx[k]), b = fmaxf(mn[k], mx[k]);
            ele[6 * (size_t)p + k] = a; ele[6 * (size_t)p + 3 + k] = b;
            gmn[k] = fminf(gmn[k], a); gmx[k] = fmaxf(gmx[k], b);
        }
    }
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = wave_min(gmn[k]), b = wave_max(gmx[k]);
        if (lane_id() == 0) { s_mn[wave][k] = a; s_mx[wave][k] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = fminf(fminf(s_mn[0][k], s_mn[1][k]), fminf(s_mn[2][k], s_mn[3][k])), b = fmaxf(fmaxf(s_mx[0][k], s_mx[1][k]), fmaxf(s_mx[2][k], s_mx[3][k]));
        atomicMin(&extent[k], f2ord(a)); atomicMax(&extent[3 + k], f2ord(b));
    }
}

MR_DEV uint32_t expand_bits(uint32_t v) {  // lbvh_morton_codes.slang:24-31
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// morton_codes (lbvh_morton_codes.slang:46-80)
__global__ void __launch_bounds__(256) k_morton(const float* __restrict__ ele, const uint32_t* __restrict__ extent, int T,
                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T) return;
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float gmin = ord2f(extent[k]), gmax = ord2f(extent[3 + k]);
        float mn = ele[6 * (size_t)g + k], mx = ele[6 * (size_t)g + 3 + k];
        float center = mn + 0.5f * (mx - mn);
        float m = (center - gmin) / (gmax - gmin);
        m = fminf(fmaxf(m * 1024.0f, 0.0f), 1023.0f);
        q[k] = expand_bits((uint32_t)m);
    }
    keys[g] = q[0] * 4 + q[1] * 2 + q[2];
    vals[g] = (uint32_t)g;
}

// ---------------------------------------------------------------- a-3: stable radix sort of (Morton code, element) pairs (lbvh_single_radixsort.slang)
// The reference sorts with ONE 256-thread workgroup (8-bit digits, least significant first, 32-wide subgroup ballots). The contract is the order: ascending
// codes, equal codes in ascending element order (the hierarchy's delta() breaks ties by position, so stability is part of the node arrays). Here: four
// 8-bit passes, each  histogram per 2048-key tile -> per digit: exclusive scan over the tiles + total -> stable scatter.  Inside a tile a wave owns 512 consecutive keys
// and takes them 64 at a time: the lanes holding the same digit find each other with eight ballots (wave64 match), the lowest of them bumps the wave's own
// LDS counter of that digit, and a key's rank is  tile base of its digit + its digit's count in the waves before + in this wave's earlier rounds + in lower lanes.
#define MR_RS_BLOCK 256
#define MR_RS_ITEMS 8
#define MR_RS_TILE (MR_RS_BLOCK * MR_RS_ITEMS)
__global__ void __launch_bounds__(MR_RS_BLOCK) k_rs_hist(const uint32_t* __restrict__ keys, int n, int shift, uint32_t* __restrict__ hist, int nb) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const int base = blockIdx.x * MR_RS_TILE;
#pragma unroll
    for (int i = 0; i < MR_RS_ITEMS; i++) {
        const int idx = base + i * MR_RS_BLOCK + (int)threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];      // digit-major: the scan below yields, per digit, the tiles in order
}
// per digit (one workgroup each): exclusive scan of its tile counters in place + the digit's total; the scatter kernel adds the totals of the smaller digits
__global__ void __launch_bounds__(256) k_rs_scan(uint32_t* __restrict__ hist, int nb, uint32_t* __restrict__ totals) {
    __shared__ uint32_t wsum[4];
    uint32_t* h = hist + (size_t)blockIdx.x * nb;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    uint32_t running = 0u;
    for (int c0 = 0; c0 < nb; c0 += 256) {
        const int i = c0 + (int)threadIdx.x;
        const uint32_t x = i < nb ? h[i] : 0u;
        uint32_t inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = 0u, all = 0u;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t t = wsum[w]; if (w < wave) before += t; all += t; }
        if (i < nb) h[i] = running + before + inc - x;
        running += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = running;
}
__global__ void __launch_bounds__(MR_RS_BLOCK) k_rs_scatter(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t* __restrict__ kout,
                                                            uint32_t* __restrict__ vout, int n, int shift, const uint32_t* __restrict__ offs, int nb,
                                                            const uint32_t* __restrict__ totals) {
    __shared__ uint32_t wh[MR_RS_BLOCK / 64][256];
    __shared__ uint32_t dsum[MR_RS_BLOCK / 64];
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int w = 0; w < MR_RS_BLOCK / 64; w++) wh[w][threadIdx.x] = 0u;
    __syncthreads();
    const int base = blockIdx.x * MR_RS_TILE + wave * (64 * MR_RS_ITEMS);
    uint32_t k[MR_RS_ITEMS], v[MR_RS_ITEMS], r[MR_RS_ITEMS];
#pragma unroll
    for (int i = 0; i < MR_RS_ITEMS; i++) {
        const int idx = base + i * 64 + lane;
        const bool valid = idx < n;
        k[i] = valid ? kin[idx] : 0xffffffffu; v[i] = valid ? vin[idx] : 0u;
        const uint32_t d = (k[i] >> shift) & 255u;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) { const bool bit = (d >> b) & 1u; const uint64_t bal = __ballot(bit); m &= bit ? bal : ~bal; }
        // m = the valid lanes of this round with the same digit (this lane included when it is valid)
        const int leader = valid ? __builtin_ctzll(m) : lane;
        uint32_t prev = 0u;
        if (valid && lane == leader) { prev = wh[wave][d]; wh[wave][d] = prev + (uint32_t)__popcll(m); }
        prev = (uint32_t)__shfl((int)prev, leader, 64);
        r[i] = prev + (uint32_t)__popcll(m & lt_mask);
        __syncthreads();                                                  // the next round's leaders read what this round's leaders wrote
    }
    {   // per digit (thread d = digit d): where each wave's keys start in the output = all keys with smaller digits + this digit's keys in the tiles before
        // + its counts in the waves before
        const uint32_t tot = totals[threadIdx.x];
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) dsum[wave] = inc;
        __syncthreads();
        uint32_t run = inc - tot + offs[(size_t)threadIdx.x * nb + blockIdx.x];
        for (int w = 0; w < wave; w++) run += dsum[w];
#pragma unroll
        for (int w = 0; w < MR_RS_BLOCK / 64; w++) { const uint32_t t = wh[w][threadIdx.x]; wh[w][threadIdx.x] = run; run += t; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MR_RS_ITEMS; i++) {
        if (base + i * 64 + lane < n) {
            const uint32_t pos = wh[wave][(k[i] >> shift) & 255u] + r[i];
            kout[pos] = k[i]; vout[pos] = v[i];
        }
    }
}
// keys / vals hold the input and receive the sorted pairs (four passes: back in place); tmp_k / tmp_v = the other half of the ping-pong, hist = 256 * tiles + 256 words
static void radix_sort_pairs_u32(uint32_t* keys, uint32_t* vals, uint32_t* tmp_k, uint32_t* tmp_v, uint32_t* hist, int n, hipStream_t s) {
    const int nb = (n + MR_RS_TILE - 1) / MR_RS_TILE;
    uint32_t *ka = keys, *va = vals, *kb = tmp_k, *vb = tmp_v;
    for (int shift = 0; shift < 32; shift += 8) {
        k_rs_hist<<<nb, MR_RS_BLOCK, 0, s>>>(ka, n, shift, hist, nb);
        k_rs_scan<<<256, 256, 0, s>>>(hist, nb, hist + (size_t)256 * nb);
        k_rs_scatter<<<nb, MR_RS_BLOCK, 0, s>>>(ka, va, kb, vb, n, shift, hist, nb, hist + (size_t)256 * nb);
        uint32_t* t = ka; ka = kb; kb = t; t = va; va = vb; vb = t;
    }
}

// delta / determineRange / findSplit (lbvh_hierarchy.slang:40-109)
MR_DEV int delta(int i, uint32_t codeI, int j, int n, const uint32_t* __restrict__ codes) {
    if (j < 0 || j > n - 1) return -1;
    uint32_t codeJ = codes[j];
    if (codeI == codeJ) return 32 + __clz((uint32_t)i ^ (uint32_t)j);  // 32 + 31 - msb ; i != j here
    return __clz(codeI ^ codeJ);                                         // 31 - msb
}

// hierarchy (lbvh_hierarchy.slang:111-245)
__global__ void __launch_bounds__(256) k_hierarchy(int T, const uint32_t* __restrict__ codes, const uint32_t* __restrict__ elem,
                                                   const float* __restrict__ ele, int32_t* __restrict__ info, float* __restrict__ aabb,
                                                   int32_t* __restrict__ parent, uint32_t* __restrict__ flags, int32_t* __restrict__ sorted_out) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T) return;
    const int LEAF = T - 1;
    {
        uint32_t e = elem[g];
        size_t n = (size_t)LEAF + g;
        info[3 * n] = 0; info[3 * n + 1] = 0; info[3 * n + 2] = (int32_t)e;
#pragma unroll
        for (int k = 0; k < 6; k++) aabb[6 * n + k] = ele[6 * (size_t)e + k];
        if (sorted_out) { sorted_out[2 * g] = (int32_t)codes[g]; sorted_out[2 * g + 1] = (int32_t)e; }
    }
    if (g < T - 1) {
        uint32_t code = codes[g];
        int dL = delta(g, code, g - 1, T, codes), dR = delta(g, code, g + 1, T, codes);
        int d = (dR >= dL) ? 1 : -1;
        int dMin = min(dL, dR);
        int lMax = 2;
        while (delta(g, code, g + lMax * d, T, codes) > dMin) lMax <<= 1;
        int l = 0;
        for (int t = lMax >> 1; t > 0; t >>= 1)
            if (delta(g, code, g + (l + t) * d, T, codes) > dMin) l += t;
        int j = g + l * d;
        int first = min(g, j), last = max(g, j);
        uint32_t firstCode = codes[first];
        int common = delta(first, firstCode, last, T, codes);
        int split = first, stride = last - first;
        do {
            stride = (stride + 1) >> 1;
            int ns = split + stride;
            if (ns < last) {
                int sp = delta(first, firstCode, ns, T, codes);
                if (sp > common) split = ns;
            }
        } while (stride > 1);
        int cA = (split == first) ? LEAF + split : split;
        int cB = (split + 1 == last) ? LEAF + split + 1 : split + 1;
        info[3 * (size_t)g] = cA; info[3 * (size_t)g + 1] = cB; info[3 * (size_t)g + 2] = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { aabb[6 * (size_t)g + k] = 1e9f; aabb[6 * (size_t)g + 3 + k] = -1e9f; }
        parent[cA] = g; parent[cB] = g;
        flags[2 * (size_t)g] = (uint32_t)first; flags[2 * (size_t)g + 1] = (uint32_t)last;
    }
    if (g == 0) parent[0] = 0;
}

// Refit: same boxes as get_bbox x tree_height + set_root (lbvh_bounding_boxes.slang:151-389). An LBVH node covers a contiguous range
// [first, last] of the sorted leaves and its box is the fmin / fmax union of their boxes — exact and order-free — so every node can be
// computed on its own from a 64-ary union pyramid over the sorted leaf boxes: no arrival counters, no device-scope fences (the bottom-up
// version with agent-scope acq_rel counters took 1.06 ms for 336 k triangles: every release writes the XCD's L2 back), no launch per level.
// level 0 = the leaf boxes aabb[T-1 ..]; level l+1 entry e = union of level-l entries [64 e, 64 e + 63].
__global__ void __launch_bounds__(256) k_refit_level(int n_src, const float* __restrict__ src, float* __restrict__ dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_dst = (n_src + 63) >> 6;
    if (e >= n_dst) return;
    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const int i1 = min(64 * e + 64, n_src);
    for (int i = 64 * e; i < i1; i++) {
#pragma unroll
        for (int k = 0; k < 3; k++) { b[k] = fminf(b[k], src[6 * (size_t)i + k]); b[3 + k] = fmaxf(b[3 + k], src[6 * (size_t)i + 3 + k]); }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) dst[6 * (size_t)e + k] = b[k];
}
struct RefitLevels { const float* a[5]; int n; };
__global__ void __launch_bounds__(256) k_refit_ranges(int T, const uint32_t* __restrict__ range, RefitLevels Lv, float* __restrict__ aabb) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T - 1) return;
    uint32_t lo = range[2 * (size_t)g], hi = range[2 * (size_t)g + 1] + 1u;   // [lo, hi) at the current level
    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int l = 0; l < Lv.n; l++) {
        const float* __restrict__ A = Lv.a[l];
        auto take = [&](uint32_t i) {
#pragma unroll
            for (int k = 0; k < 3; k++) { b[k] = fminf(b[k], A[6 * (size_t)i + k]); b[3 + k] = fmaxf(b[3 + k], A[6 * (size_t)i + 3 + k]); }
        };
        while (lo < hi && (lo & 63u)) take(lo++);
        while (lo < hi && (hi & 63u)) take(--hi);
        if (lo >= hi) break;
        if (l + 1 == Lv.n) { for (uint32_t i = lo; i < hi; i++) take(i); break; }   // top level: fewer than 64 entries
        lo >>= 6; hi >>= 6;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) aabb[6 * (size_t)g + k] = b[k];
}

// Pack the traversal layout: 64-byte two-child records + per-leaf triangle records (v0, e1, e2 pre-subtracted:
// the same fp32 subtraction triangle_hit performs, helperDi.slang:175-176).
__global__ void __launch_bounds__(256) k_pack(int T, const int32_t* __restrict__ info, const float* __restrict__ aabb,
                                              const float* __restrict__ vert, const int32_t* __restrict__ tri, WideNode* __restrict__ nodes,
                                              TriRec* __restrict__ tris, float* __restrict__ root_box) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T) return;
    const int LEAF = T - 1;
    {
        int prim = info[3 * ((size_t)LEAF + g) + 2];
        const int32_t* ti = tri + 3 * (size_t)prim;
        v3 a = ld3(vert, ti[0]), b = ld3(vert, ti[1]), c = ld3(vert, ti[2]);
        v3 e1 = b - a, e2 = c - a;
        TriRec r;
        r.v0[0] = a.x; r.v0[1] = a.y; r.v0[2] = a.z; r.e1[0] = e1.x; r.e1[1] = e1.y; r.e1[2] = e1.z; r.e2[0] = e2.x; r.e2[1] = e2.y; r.e2[2] = e2.z;
        r.prim = prim; r.pad[0] = 0; r.pad[1] = 0;
        tris[g] = r;
    }
    if (g < T - 1) {
        int L = info[3 * (size_t)g], R = info[3 * (size_t)g + 1];
        WideNode n;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            n.lmin[k] = aabb[6 * (size_t)L + k]; n.lmax[k] = aabb[6 * (size_t)L + 3 + k];
            n.rmin[k] = aabb[6 * (size_t)R + k]; n.rmax[k] = aabb[6 * (size_t)R + 3 + k];
        }
        n.left = (L >= LEAF) ? ~(L - LEAF) : L;
        n.right = (R >= LEAF) ? ~(R - LEAF) : R;
        n.pad[0] = 0; n.pad[1] = 0;
        nodes[g] = n;
    }
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) root_box[k] = aabb[k];
    }
}

// ---- compressed 4-wide layout for shadow rays (engine.hpp Node4q / LeafRec)
// step 2^(E-127) per axis: the smallest power of two (E >= 67) for which the decode expression of q = 255 still reaches the node's max corner
MR_DEV float q_step(uint32_t E) { return __uint_as_float(E << 23); }
MR_DEV float q_decode(uint32_t q, float step, float org) { return fmaf((float)q, step, org); }   // q*step is exact => one rounding, same value as mul+add
// `info` / `aabb`: the hierarchy the 4-wide nodes are collapsed from — the reference LBVH itself, or (PRIV) the private steering hierarchy over the same
// leaves (k_emc_* below), whose leaf entries carry the reference SLOT in info[.][2]; `rinfo` / `raabb`: always the reference arrays (leaf records).
template <bool PRIV>
__global__ void __launch_bounds__(256) k_pack4q(int T, const int32_t* __restrict__ info, const float* __restrict__ aabb, const int32_t* __restrict__ rinfo,
                                                const float* __restrict__ raabb, const float* __restrict__ vert,
                                                const int32_t* __restrict__ tri, Node4q* __restrict__ nodes4q, LeafRec* __restrict__ leaves) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T) return;
    const int LEAF = T - 1;
    {
        const int prim = rinfo[3 * ((size_t)LEAF + g) + 2];
        const int32_t* ti = tri + 3 * (size_t)prim;
        v3 a = ld3(vert, ti[0]), b = ld3(vert, ti[1]), c = ld3(vert, ti[2]);
        v3 e1 = b - a, e2 = c - a;
        const float* bx = raabb + 6 * ((size_t)LEAF + g);
        LeafRec r;
        r.v0[0] = a.x; r.v0[1] = a.y; r.v0[2] = a.z; r.e1[0] = e1.x; r.e1[1] = e1.y; r.e1[2] = e1.z; r.e2[0] = e2.x; r.e2[1] = e2.y; r.e2[2] = e2.z;
#pragma unroll
        for (int k = 0; k < 3; k++) { r.lo[k] = bx[k]; r.hi[k] = bx[3 + k]; }
        r.prim = prim;
        leaves[g] = r;
        if (g == 0) {   // leaves[T]: the "null leaf" every unused child slot refers to — an inverted box no ray passes, so the traversal kernels need no
                        // test for unused slots (their decoded boxes fail by themselves; should one ever pass, this record's exact test rejects it)
            LeafRec z;
#pragma unroll
            for (int k = 0; k < 3; k++) { z.v0[k] = 0.f; z.e1[k] = 0.f; z.e2[k] = 0.f; z.lo[k] = 3.0e38f; z.hi[k] = -3.0e38f; }
            z.prim = -1;
            leaves[T] = z;
        }
    }
    if (g >= T - 1) return;
    // children: start from the node's two LBVH children and keep opening the internal entry with the largest surface area until there are
    // four (the usual wide-BVH collapse; any hierarchy over the same leaves gives the same any-hit bit). Entries stay LBVH node ids.
    int c[4]; int nc = 2;
    c[0] = info[3 * (size_t)g]; c[1] = info[3 * (size_t)g + 1];
    for (int round = 0; round < 2; round++) {
        int pick = -1; float best = -1.f;
        for (int k = 0; k < nc; k++) {
            if (c[k] >= LEAF) continue;
            const float* b = aabb + 6 * (size_t)c[k];
            const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > best) { best = area; pick = k; }
        }
        if (pick < 0) break;
        const int open = c[pick];
        c[pick] = info[3 * (size_t)open]; c[nc++] = info[3 * (size_t)open + 1];
    }
    float cb[4][6];
    for (int k = 0; k < nc; k++) for (int a = 0; a < 6; a++) cb[k][a] = aabb[6 * (size_t)c[k] + a];
    Node4q n;
    n.step_x = n.step_y = n.step_z = 0.f;
    for (int a = 0; a < 3; a++) {
        float lo = cb[0][a], hi = cb[0][3 + a];
        for (int k = 1; k < nc; k++) { lo = fminf(lo, cb[k][a]); hi = fmaxf(hi, cb[k][3 + a]); }
        const float r = (hi - lo) / 255.0f;
        uint32_t E = ((__float_as_uint(r) >> 23) & 0xffu);
        if (E < 67u) E = 67u;
        if (E > 254u) E = 254u;
        while (E < 254u && q_decode(255u, q_step(E), lo) < hi) E++;
        const float st = q_step(E);
        uint32_t wlo = 0, whi = 0;
        for (int k = 0; k < 4; k++) {
            uint32_t ql = 255u, qh = 0u;   // unused entry: entry beyond exit by the whole node extent; refers to the null leaf
            if (k < nc) {
                const float fl = fminf(fmaxf(floorf((cb[k][a] - lo) / st), 0.f), 255.f);
                const float fh = fminf(fmaxf(ceilf((cb[k][3 + a] - lo) / st), 0.f), 255.f);
                ql = (uint32_t)fl; qh = (uint32_t)fh;
                while (ql > 0u && q_decode(ql, st, lo) > cb[k][a]) ql--;                 // outward: decoded min <= exact min (q = 0 decodes to lo itself)
                while (ql < 255u && q_decode(ql + 1u, st, lo) <= cb[k][a]) ql++;         // and as tight as the grid allows
                while (qh < 255u && q_decode(qh, st, lo) < cb[k][3 + a]) qh++;           // decoded max >= exact max (q = 255 reaches hi by the choice of E)
                while (qh > 0u && q_decode(qh - 1u, st, lo) >= cb[k][3 + a]) qh--;
            }
            wlo |= ql << (8 * k); whi |= qh << (8 * k);
        }
        n.org[a] = lo; (a == 0 ? n.step_x : a == 1 ? n.step_y : n.step_z) = st; n.qlo[a] = wlo; n.qhi[a] = whi;
    }
    for (int k = 0; k < 4; k++) n.ref[k] = (k < nc) ? ((c[k] >= LEAF) ? ~(PRIV ? info[3 * (size_t)c[k] + 2] : (c[k] - LEAF)) : c[k]) : ~T;      // ~T: the null leaf (leaves[T])
    nodes4q[g] = n;
}

// ---------------------------------------------------------------- the private steering hierarchy (round 4): extended Morton codes
// The shadow-ray bit and the ordered closest hit depend on the LEAVES' own boxes only (bvh_trace.hip), so the hierarchy the 4-wide nodes are collapsed
// from is free. On meshes with very different triangle sizes (the lego-like scene: areas spread 10^5 : 1) the reference LBVH lets every large triangle
// inflate the boxes of the small ones sorted next to it. Extended Morton codes (Vinkler, Bittner, Havran 2017) make the triangle's SIZE a fourth coordinate:
// key = 8 levels of (size bit, x bit, y bit, z bit), size = box diagonal / scene diagonal in 8 linear bits — large triangles split off in the top levels.
// The leaves are re-sorted by that 32-bit key with the same stable radix sort, starting from the reference (Morton) order, so equal keys keep their Morton
// order; the hierarchy is Karras' over the 64-bit key (extended code, 30-bit Morton code), ties by position. Measured on the replay of this kernel's visiting
// order (scripts/treelab): lego-like mesh 16.6 -> 14.8 records per shadow ray (-11 %), 59.8 -> 52.3 box tests; uniformly tessellated icosphere: unchanged
// (all its triangles have size bits 0: the same order, the same tree). A PLOC hierarchy does another 4 % better and costs ~25 more launches per build.
// Depth: at most 32 + 6 (the Morton bits the extended code does not already hold) + ceil(log2 T) levels -> the private stack bound of bvh_trace.hip.
MR_DEV uint32_t spread4(uint32_t v) {   // bit i of an 8-bit value -> bit 4 i
    v &= 0xffu; v = (v | (v << 12)) & 0x000f000fu; v = (v | (v << 6)) & 0x03030303u; v = (v | (v << 3)) & 0x11111111u;
    return v;
}
__global__ void __launch_bounds__(256) k_emc_keys(int T, const float* __restrict__ aabb, const uint32_t* __restrict__ extent, uint32_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;      // reference slot
    if (g >= T) return;
    const float* b = aabb + 6 * ((size_t)(T - 1) + g);
    float sd = 0.f, dg = 0.f; uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float gmin = ord2f(extent[k]), gmax = ord2f(extent[3 + k]);
        const float e = gmax - gmin, d = b[3 + k] - b[k];
        sd += e * e; dg += d * d;
        const float m = ((b[k] + 0.5f * d) - gmin) / e;
        q[k] = (uint32_t)fminf(fmaxf(m * 256.0f, 0.0f), 255.0f);
    }
    const uint32_t sz = (uint32_t)fminf(fmaxf(sqrtf(dg) / sqrtf(sd) * 256.0f, 0.0f), 255.0f);
    keys[g] = (spread4(sz) << 3) | (spread4(q[0]) << 2) | (spread4(q[1]) << 1) | spread4(q[2]);
    vals[g] = (uint32_t)g;
}
// after the sort: 64-bit keys, the leaf level of the private arrays (box of the slot, info = (0, 0, slot))
__global__ void __launch_bounds__(256) k_emc_leaves(int T, const uint32_t* __restrict__ ekeys, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ codes,
                                                    const float* __restrict__ aabb, unsigned long long* __restrict__ key64, int32_t* __restrict__ pinfo,
                                                    float* __restrict__ paabb) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;      // position in the private order
    if (j >= T) return;
    const uint32_t slot = slots[j];
    key64[j] = ((unsigned long long)ekeys[j] << 32) | (unsigned long long)codes[slot];
    const size_t n = (size_t)(T - 1) + j;
    pinfo[3 * n] = 0; pinfo[3 * n + 1] = 0; pinfo[3 * n + 2] = (int32_t)slot;
#pragma unroll
    for (int k = 0; k < 6; k++) paabb[6 * n + k] = aabb[6 * ((size_t)(T - 1) + slot) + k];
}
MR_DEV int delta64(int i, unsigned long long ki, int j, int n, const unsigned long long* __restrict__ keys) {
    if (j < 0 || j > n - 1) return -1;
    const unsigned long long kj = keys[j];
    if (ki == kj) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll(ki ^ kj);
}
__global__ void __launch_bounds__(256) k_hierarchy64(int T, const unsigned long long* __restrict__ keys, int32_t* __restrict__ info, uint32_t* __restrict__ range, int32_t* __restrict__ parent) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T - 1) return;
    const int LEAF = T - 1;
    const unsigned long long key = keys[g];
    const int dL = delta64(g, key, g - 1, T, keys), dR = delta64(g, key, g + 1, T, keys);
    const int d = (dR >= dL) ? 1 : -1;
    const int dMin = min(dL, dR);
    int lMax = 2;
    while (delta64(g, key, g + lMax * d, T, keys) > dMin) lMax <<= 1;
    int l = 0;
    for (int t = lMax >> 1; t > 0; t >>= 1)
        if (delta64(g, key, g + (l + t) * d, T, keys) > dMin) l += t;
    const int j = g + l * d;
    const int first = min(g, j), last = max(g, j);
    const unsigned long long firstKey = keys[first];
    const int common = delta64(first, firstKey, last, T, keys);
    int split = first, stride = last - first;
    do {
        stride = (stride + 1) >> 1;
        const int ns = split + stride;
        if (ns < last && delta64(first, firstKey, ns, T, keys) > common) split = ns;
    } while (stride > 1);
    const int cA = (split == first) ? LEAF + split : split, cB = (split + 1 == last) ? LEAF + split + 1 : split + 1;
    info[3 * (size_t)g] = cA; info[3 * (size_t)g + 1] = cB; info[3 * (size_t)g + 2] = 0;
    range[2 * (size_t)g] = (uint32_t)first; range[2 * (size_t)g + 1] = (uint32_t)last;
    parent[cA] = g; parent[cB] = g;
    if (g == 0) parent[0] = -1;
}

// ---------------------------------------------------------------- SAH top over prefix clusters (round 4; the default, MIRRES_PRIVATE_TREE=1 keeps the plain extended-Morton tree)
// The upper levels of the private hierarchy rebuilt top-down with a binned surface-area heuristic. Items = the maximal subtrees of the extended-Morton tree whose
// leaves share a key prefix of MR_SAH_PREFIX bits (six (size, x, y, z) levels: ~10-20 k clusters for 3.3e5 triangles); everything above them is replaced: per level
// every node bins its items' centroids into 8 bins per axis (box + count per bin, device atomics), picks the cheapest of the 21 candidate planes, and its items move
// to the two children. The rebuilt nodes take over the ids of the nodes they replace (a cut through a binary tree with C subtrees below it has C - 1 nodes above it),
// node 0 stays the root. scripts/treelab (CPU replay of the shadow-ray kernel's visiting order): records per ray -6 % on both meshes beyond the extended-Morton tree;
// measured (profiles/r04_ab_sah_top.txt): shadow-ray launch -4 % (icosphere) / -7 % (lego-like), 128-spp frame +2.5 % / +3.8 %, build +0.9 / +1.4 ms.
// All or nothing: if anything is off — more clusters than the scratch holds, a level budget exceeded, counts that do not add up — nothing is installed and the
// extended-Morton tree stays as it is (k_sah_install checks). Which ids the atomics hand out varies from run to run; the topology does not, and no result depends on either.
#define MR_SAH_BINS 8
#define MR_SAH_MAXC 65536
struct SahNode { uint32_t count, cb[6], minidx; int32_t axis, bin; float lo, scale; int32_t cl, cr, iid, alias; float box[6]; };      // 24 words
struct SahBin { uint32_t count, box[6]; };
struct SahState { uint32_t n_items, n_top, n_nodes, n_internal, n_resolved, fail, first[MR_SAH_LEVELS + 3]; };
MR_DEV int node_prefix(int n, int T, const unsigned long long* __restrict__ key64, const uint32_t* __restrict__ range) {
    if (n >= T - 1) return 128;                                  // a leaf
    const unsigned long long a = key64[range[2 * (size_t)n]], b = key64[range[2 * (size_t)n + 1]];
    return a == b ? 64 : __clzll(a ^ b);
}
MR_DEV void sah_accumulate(SahNode* nodes, int k, const float* __restrict__ bx, uint32_t item) {
    SahNode& N = nodes[k];
    atomicAdd(&N.count, 1u); atomicMin(&N.minidx, item);
#pragma unroll
    for (int a = 0; a < 3; a++) { const uint32_t c = f2ord(bx[a] + 0.5f * (bx[3 + a] - bx[a])); atomicMin(&N.cb[a], c); atomicMax(&N.cb[3 + a], c); }
}
__global__ void __launch_bounds__(256) k_sah_reset(SahState* st, SahNode* nodes, int n_nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { st->n_items = 0; st->n_top = 1; st->n_nodes = 1; st->n_internal = 1; st->n_resolved = 0; st->fail = 0; st->first[0] = 0; st->first[1] = 1; st->first[2] = 1; }
    if (i < n_nodes) { SahNode z; z.count = 0; z.minidx = 0xffffffffu; for (int a = 0; a < 3; a++) { z.cb[a] = 0xffffffffu; z.cb[3 + a] = 0u; } z.axis = -1; z.bin = 0; z.lo = 0.f; z.scale = 0.f; z.cl = z.cr = -1; z.iid = -1; z.alias = -1;
                       for (int a = 0; a < 6; a++) z.box[a] = 0.f; nodes[i] = z; }
}
__global__ void __launch_bounds__(256) k_sah_clusters(int T, const unsigned long long* __restrict__ key64, const uint32_t* __restrict__ range, const int32_t* __restrict__ parent,
                                                      const float* __restrict__ paabb, SahState* st, SahNode* nodes, int32_t* __restrict__ iref, int32_t* __restrict__ inode,
                                                      int32_t* __restrict__ top_ids) {
    // the root's item count and centroid bounds: one set of atomics per workgroup (every cluster would otherwise hit the same eight words)
    __shared__ uint32_t s_count, s_min, s_cb[6];
    if (threadIdx.x == 0) { s_count = 0; s_min = 0xffffffffu; for (int a = 0; a < 3; a++) { s_cb[a] = 0xffffffffu; s_cb[3 + a] = 0u; } }
    __syncthreads();
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < 2 * T - 1; n += gridDim.x * blockDim.x) {
        const int cp = node_prefix(n, T, key64, range);
        const int par = parent[n];
        const int cpp = par < 0 ? -1 : node_prefix(par, T, key64, range);
        if (cp >= MR_SAH_PREFIX && cpp < MR_SAH_PREFIX) {            // a cluster root (n == 0 here: the whole tree is one cluster, nothing to rebuild)
            const uint32_t i = atomicAdd(&st->n_items, 1u);
            if (i < MR_SAH_MAXC) {
                iref[i] = n; inode[i] = 0;
                const float* bx = paabb + 6 * (size_t)n;
                atomicAdd(&s_count, 1u); atomicMin(&s_min, i);
#pragma unroll
                for (int a = 0; a < 3; a++) { const uint32_t c = f2ord(bx[a] + 0.5f * (bx[3 + a] - bx[a])); atomicMin(&s_cb[a], c); atomicMax(&s_cb[3 + a], c); }
            } else st->fail = 1;
        } else if (cp < MR_SAH_PREFIX) {                             // an internal node above the cut: its id is reused
            if (n == 0) top_ids[0] = 0; else { const uint32_t j = atomicAdd(&st->n_top, 1u); if (j < MR_SAH_MAXC) top_ids[j] = n; else st->fail = 1; }
        }
        if (n == 0) { for (int a = 0; a < 6; a++) nodes[0].box[a] = paabb[a]; nodes[0].iid = 0; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_count) {
        atomicAdd(&nodes[0].count, s_count); atomicMin(&nodes[0].minidx, s_min);
        for (int a = 0; a < 3; a++) { atomicMin(&nodes[0].cb[a], s_cb[a]); atomicMax(&nodes[0].cb[3 + a], s_cb[3 + a]); }
    }
}
MR_DEV int sah_bin_of(float c, float lo, float scale) { const int b = (int)((c - lo) * scale); return b < 0 ? 0 : (b > MR_SAH_BINS - 1 ? MR_SAH_BINS - 1 : b); }
// Levels 0 .. MR_SAH_LDS_LEVELS - 1 have at most 32 nodes (64 children): thousands of items would hammer the same few words with device atomics (the first version
// spent 4.4 of its 4.9 ms there), so every workgroup first accumulates in LDS and then sends one atomic per non-empty word. Deeper levels go straight to memory.
#define MR_SAH_LDS_LEVELS 6
#define MR_SAH_LDS_NODES 32
// levels from MR_SAH_TAIL on are finished by ONE workgroup looping over the remaining levels (the trees of both test meshes are complete by level 18 / 21: the
// tail usually finds nothing to do; started at level 14 it cost 2.4 ms on the lego-like mesh — a third of the items were still unplaced): k_sah_tail
#define MR_SAH_TAIL 22
MR_DEV void sah_bin_item(uint32_t i, int f0, const SahNode* __restrict__ nodes, const int32_t* __restrict__ iref, const int32_t* __restrict__ inode,
                         const float* __restrict__ paabb, SahBin* bins) {      // bins: global, or the workgroup's LDS copy (same indexing)
    const int k = inode[i];
    if (k < f0) return;                                      // resolved (-1) — every live item sits in a node of the current level
    const SahNode& N = nodes[k];
    if (N.count < 2) return;
    const float* bx = paabb + 6 * (size_t)iref[i];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = ord2f(N.cb[a]), hi = ord2f(N.cb[3 + a]);
        if (!(hi > lo)) continue;
        const int b = sah_bin_of(bx[a] + 0.5f * (bx[3 + a] - bx[a]), lo, (float)MR_SAH_BINS / (hi - lo));
        SahBin& B = bins[((size_t)(k - f0) * 3 + a) * MR_SAH_BINS + b];
        atomicAdd(&B.count, 1u);
#pragma unroll
        for (int q = 0; q < 3; q++) { atomicMin(&B.box[q], f2ord(bx[q])); atomicMax(&B.box[3 + q], f2ord(bx[3 + q])); }
    }
}
template <bool LDS>
__global__ void __launch_bounds__(256) k_sah_bin(int level, const SahState* st, const SahNode* __restrict__ nodes, const int32_t* __restrict__ iref, const int32_t* __restrict__ inode,
                                                 const float* __restrict__ paabb, SahBin* __restrict__ bins) {
    __shared__ SahBin sb[LDS ? MR_SAH_LDS_NODES * 3 * MR_SAH_BINS : 1];
    const uint32_t C = st->n_items < MR_SAH_MAXC ? st->n_items : MR_SAH_MAXC;
    const int f0 = (int)st->first[level];
    if (LDS) {
        for (int j = threadIdx.x; j < MR_SAH_LDS_NODES * 3 * MR_SAH_BINS; j += blockDim.x) { sb[j].count = 0; for (int q = 0; q < 3; q++) { sb[j].box[q] = 0xffffffffu; sb[j].box[3 + q] = 0u; } }
        __syncthreads();
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x) sah_bin_item(i, f0, nodes, iref, inode, paabb, LDS ? sb : bins);
    if (LDS) {
        __syncthreads();
        for (int j = threadIdx.x; j < MR_SAH_LDS_NODES * 3 * MR_SAH_BINS; j += blockDim.x) {
            if (!sb[j].count) continue;
            atomicAdd(&bins[j].count, sb[j].count);
            for (int q = 0; q < 3; q++) { atomicMin(&bins[j].box[q], sb[j].box[q]); atomicMax(&bins[j].box[3 + q], sb[j].box[3 + q]); }
        }
    }
}
MR_DEV void sah_split_node(int k, int level, int f0, SahState* st, SahNode* nodes, SahBin* __restrict__ bins) {
    SahNode& N = nodes[k];
    if (N.count < 2) return;                                     // a single item: resolved by the assign step (count 0 cannot happen: both sides of a split are non-empty)
    if (level >= MR_SAH_LEVELS) { st->fail = 1; return; }
    float best = 3.0e38f; int best_axis = -1, best_bin = 0; uint32_t best_nl = 0; float lbox[6], rbox[6];
    for (int a = 0; a < 3; a++) {
        SahBin* B = bins + ((size_t)(k - f0) * 3 + a) * MR_SAH_BINS;
        const float lo = ord2f(N.cb[a]), hi = ord2f(N.cb[3 + a]);
        if (hi > lo) {
            float racc[MR_SAH_BINS][6]; uint32_t rcnt[MR_SAH_BINS];
            float acc[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}; uint32_t c = 0;
            for (int b = MR_SAH_BINS - 1; b >= 0; b--) {
                if (B[b].count) { for (int q = 0; q < 3; q++) { acc[q] = fminf(acc[q], ord2f(B[b].box[q])); acc[3 + q] = fmaxf(acc[3 + q], ord2f(B[b].box[3 + q])); } c += B[b].count; }
                for (int q = 0; q < 6; q++) racc[b][q] = acc[q]; rcnt[b] = c;
            }
            float l[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}; uint32_t nl = 0;
            for (int b = 0; b + 1 < MR_SAH_BINS; b++) {
                if (B[b].count) { for (int q = 0; q < 3; q++) { l[q] = fminf(l[q], ord2f(B[b].box[q])); l[3 + q] = fmaxf(l[3 + q], ord2f(B[b].box[3 + q])); } nl += B[b].count; }
                const uint32_t nr = rcnt[b + 1];
                if (!nl || !nr) continue;
                const float* r = racc[b + 1];
                const float al = (l[3] - l[0]) * (l[4] - l[1]) + (l[4] - l[1]) * (l[5] - l[2]) + (l[5] - l[2]) * (l[3] - l[0]);
                const float ar = (r[3] - r[0]) * (r[4] - r[1]) + (r[4] - r[1]) * (r[5] - r[2]) + (r[5] - r[2]) * (r[3] - r[0]);
                const float cost = al * (float)nl + ar * (float)nr;
                if (cost < best) { best = cost; best_axis = a; best_bin = b; best_nl = nl; for (int q = 0; q < 6; q++) { lbox[q] = l[q]; rbox[q] = r[q]; } }
            }
        }
        for (int b = 0; b < MR_SAH_BINS; b++) { B[b].count = 0; for (int q = 0; q < 3; q++) { B[b].box[q] = 0xffffffffu; B[b].box[3 + q] = 0u; } }      // clean for the next level
    }
    const uint32_t base = atomicAdd(&st->n_nodes, 2u);
    if (base + 2 > 2 * MR_SAH_MAXC) { st->fail = 1; return; }
    N.cl = (int)base; N.cr = (int)base + 1;
    if (best_axis >= 0) {
        N.axis = best_axis; N.bin = best_bin; N.lo = ord2f(N.cb[best_axis]); N.scale = (float)MR_SAH_BINS / (ord2f(N.cb[3 + best_axis]) - N.lo);
        for (int q = 0; q < 6; q++) { nodes[base].box[q] = lbox[q]; nodes[base + 1].box[q] = rbox[q]; }
        if (best_nl >= 2) nodes[base].iid = (int)atomicAdd(&st->n_internal, 1u);
        if (N.count - best_nl >= 2) nodes[base + 1].iid = (int)atomicAdd(&st->n_internal, 1u);
    } else {
        // all centroids coincide: peel off the item with the smallest index (the boxes of the two sides are formed by the assign step)
        N.axis = 3;
        for (int q = 0; q < 3; q++) { nodes[base].box[q] = nodes[base + 1].box[q] = INFINITY; nodes[base].box[3 + q] = nodes[base + 1].box[3 + q] = -INFINITY; }
        if (N.count - 1 >= 2) nodes[base + 1].iid = (int)atomicAdd(&st->n_internal, 1u);
    }
}
__global__ void __launch_bounds__(256) k_sah_split(int level, SahState* st, SahNode* nodes, SahBin* __restrict__ bins) {
    const int f0 = (int)st->first[level], f1 = (int)st->first[level + 1];
    const int k = f0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (k < f1) sah_split_node(k, level, f0, st, nodes, bins);
}
MR_DEV void atomic_box_f(float* box, const float* bx) {   // float min / max by compare-and-swap: only the few degenerate nodes come here
#pragma unroll
    for (int q = 0; q < 6; q++) {
        unsigned int* p = reinterpret_cast<unsigned int*>(box + q); unsigned int old = *p;
        while (true) { const float cur = __uint_as_float(old); const float nv = q < 3 ? fminf(cur, bx[q]) : fmaxf(cur, bx[q]); if (nv == cur) break;
                       const unsigned int prev = atomicCAS(p, old, __float_as_uint(nv)); if (prev == old) break; old = prev; }
    }
}
// per-child accumulators of the assign step: {count, centroid bounds x 6, smallest item index} — in the nodes themselves, or in a workgroup's LDS copy
struct SahAcc { uint32_t count, cb[6], minidx; };
MR_DEV void sah_assign_item(uint32_t i, int f0, int c0, SahState* st, SahNode* nodes, const int32_t* __restrict__ iref, int32_t* __restrict__ inode, const float* __restrict__ paabb, SahAcc* lacc) {
    const int k = inode[i];
    if (k < f0) return;
    SahNode& N = nodes[k];
    if (N.count == 1) { N.alias = iref[i]; inode[i] = -1; atomicAdd(&st->n_resolved, 1u); return; }
    if (N.cl < 0) return;                                    // not split (failure flagged by the split step)
    const float* bx = paabb + 6 * (size_t)iref[i];
    int side;
    if (N.axis == 3) side = (i == N.minidx) ? 0 : 1;
    else side = sah_bin_of(bx[N.axis] + 0.5f * (bx[3 + N.axis] - bx[N.axis]), N.lo, N.scale) <= N.bin ? 0 : 1;
    const int c = side ? N.cr : N.cl;
    inode[i] = c;
    if (lacc) {
        SahAcc& A = lacc[c - c0];
        atomicAdd(&A.count, 1u); atomicMin(&A.minidx, i);
#pragma unroll
        for (int a = 0; a < 3; a++) { const uint32_t cc = f2ord(bx[a] + 0.5f * (bx[3 + a] - bx[a])); atomicMin(&A.cb[a], cc); atomicMax(&A.cb[3 + a], cc); }
    } else sah_accumulate(nodes, c, bx, i);
    if (N.axis == 3) atomic_box_f(nodes[c].box, bx);
}
template <bool LDS>
__global__ void __launch_bounds__(256) k_sah_assign(int level, SahState* st, SahNode* nodes, const int32_t* __restrict__ iref, int32_t* __restrict__ inode, const float* __restrict__ paabb) {
    __shared__ SahAcc sa[LDS ? 2 * MR_SAH_LDS_NODES : 1];
    const uint32_t C = st->n_items < MR_SAH_MAXC ? st->n_items : MR_SAH_MAXC;
    const int f0 = (int)st->first[level], c0 = (int)st->first[level + 1];      // this level's nodes start at f0, their children at c0
    if (LDS) {
        for (int j = threadIdx.x; j < 2 * MR_SAH_LDS_NODES; j += blockDim.x) { sa[j].count = 0; sa[j].minidx = 0xffffffffu; for (int q = 0; q < 3; q++) { sa[j].cb[q] = 0xffffffffu; sa[j].cb[3 + q] = 0u; } }
        __syncthreads();
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x) sah_assign_item(i, f0, c0, st, nodes, iref, inode, paabb, LDS ? sa : nullptr);
    if (LDS) {
        __syncthreads();
        for (int j = threadIdx.x; j < 2 * MR_SAH_LDS_NODES; j += blockDim.x) {
            if (!sa[j].count) continue;
            SahNode& N = nodes[c0 + j];
            atomicAdd(&N.count, sa[j].count); atomicMin(&N.minidx, sa[j].minidx);
            for (int q = 0; q < 3; q++) { atomicMin(&N.cb[q], sa[j].cb[q]); atomicMax(&N.cb[3 + q], sa[j].cb[3 + q]); }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->first[level + 2] = st->n_nodes;      // no allocation runs concurrently with this kernel
}
// the remaining levels in one workgroup: bin -> split -> assign per level with workgroup barriers in between, until a level has no node left
__global__ void __launch_bounds__(1024) k_sah_tail(int level0, SahState* st, SahNode* nodes, const int32_t* __restrict__ iref, int32_t* __restrict__ inode, const float* __restrict__ paabb,
                                                   SahBin* __restrict__ bins) {
    const uint32_t C = st->n_items < MR_SAH_MAXC ? st->n_items : MR_SAH_MAXC;
    for (int level = level0; level <= MR_SAH_LEVELS; level++) {
        const int f0 = (int)st->first[level], f1 = (int)st->first[level + 1];
        if (f1 <= f0) break;                                  // uniform: every thread reads the same words after the barrier below
        for (uint32_t i = threadIdx.x; i < C; i += blockDim.x) sah_bin_item(i, f0, nodes, iref, inode, paabb, bins);
        __threadfence(); __syncthreads();
        for (int k = f0 + (int)threadIdx.x; k < f1; k += blockDim.x) sah_split_node(k, level, f0, st, nodes, bins);
        __threadfence(); __syncthreads();
        for (uint32_t i = threadIdx.x; i < C; i += blockDim.x) sah_assign_item(i, f0, f1, st, nodes, iref, inode, paabb, nullptr);
        __threadfence(); __syncthreads();
        if (threadIdx.x == 0) st->first[level + 2] = st->n_nodes;
        __threadfence(); __syncthreads();
    }
}
// the rebuilt upper levels into the private node arrays — or nothing at all when the counts do not add up
__global__ void __launch_bounds__(256) k_sah_install(const SahState* st, const SahNode* __restrict__ nodes, const int32_t* __restrict__ top_ids, int32_t* __restrict__ pinfo, float* __restrict__ paabb) {
    const uint32_t C = st->n_items;
    if (st->fail || C < 2 || C > MR_SAH_MAXC || st->n_resolved != C || st->n_internal != C - 1 || st->n_top != C - 1) return;
    const uint32_t n = st->n_nodes;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const SahNode& N = nodes[k];
        if (N.count < 2) continue;
        const int id = top_ids[N.iid];
        const SahNode &L = nodes[N.cl], &R = nodes[N.cr];
        pinfo[3 * (size_t)id] = L.count == 1 ? L.alias : top_ids[L.iid];
        pinfo[3 * (size_t)id + 1] = R.count == 1 ? R.alias : top_ids[R.iid];
        pinfo[3 * (size_t)id + 2] = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) paabb[6 * (size_t)id + q] = N.box[q];
    }
}

// Breadth-first prefix (TOPN = 85: levels 0..3, 341: levels 0..4) of the compressed 4-wide tree for the LDS-resident part of the shadow-ray
// kernel; references to children inside the prefix become MR_TOPBIT | index. Heap layout: entry e's k-th child sits at 4e + 1 + k.
__global__ void __launch_bounds__(256) k_top4q(int T, const Node4q* __restrict__ nodes4q, Node4q* __restrict__ top, int TOPN) {
    __shared__ int s_id[341];
    const int t = threadIdx.x;
    if (t == 0) s_id[0] = 0;
    __syncthreads();
    const int levels = TOPN == 85 ? 3 : 4;    // expansions needed to fill the prefix
    int first = 0, cnt = 1;
    for (int lvl = 0; lvl < levels; lvl++) {
        for (int e = first + t; e < first + cnt; e += 256) {
            const int id = s_id[e];
            for (int k = 0; k < 4; k++) {
                int r = id >= 0 ? nodes4q[id].ref[k] : -1;
                s_id[4 * e + 1 + k] = r >= 0 ? r : -1;         // leaves (the null leaf among them) are negative
            }
        }
        first += cnt; cnt *= 4;
        __syncthreads();
    }
    for (int e = t; e < TOPN; e += 256) {
        const int id = s_id[e];
        Node4q n;
        if (id >= 0) n = nodes4q[id];
        else { n.org[0] = n.org[1] = n.org[2] = 0.f; n.step_x = n.step_y = n.step_z = q_step(67u); for (int a = 0; a < 3; a++) { n.qlo[a] = 0xffffffffu; n.qhi[a] = 0u; }
               for (int k = 0; k < 4; k++) n.ref[k] = ~T; }
        for (int k = 0; k < 4; k++) {
            const int cslot = 4 * e + 1 + k;
            const bool internal = id >= 0 && n.ref[k] >= 0;
            if (internal && cslot < TOPN) n.ref[k] = 0x20000000 | cslot;
        }
        top[e] = n;
    }
}

__global__ void k_init_extent(uint32_t* extent) {
    int i = threadIdx.x;
    if (i < 3) extent[i] = 0xffffffffu; else if (i < 6) extent[i] = 0u;
}

}  // namespace mr

using namespace mr;

// ---- the private steering hierarchy in two steps (round 6: separable, so that a caller that rebuilds per frame can stop after step 1 and ask for step 2 only when the
// frame is long enough to pay for it: mirres_bvh_upgrade)
static int private_step1(mirres_bvh* b, int T, const float* aabb, hipStream_t s) {      // extended-Morton tree over the reference leaves
    const int blk = 256, grd = grid_for(T, blk);
    k_emc_keys<<<grd, blk, 0, s>>>(T, aabb, b->extent, b->p_keys, b->p_vals);
    radix_sort_pairs_u32(b->p_keys, b->p_vals, b->keys_in, b->vals_in, (uint32_t*)b->sort_tmp, T, s);      // keys_in / vals_in: idle halves of the first sort's ping-pong
    k_emc_leaves<<<grd, blk, 0, s>>>(T, b->p_keys, b->p_vals, b->keys_out, aabb, b->p_key64, b->p_info, b->p_aabb);
    k_hierarchy64<<<grd, blk, 0, s>>>(T, b->p_key64, b->p_info, b->p_range, b->p_parent);
    RefitLevels Lv; Lv.n = 1; Lv.a[0] = b->p_aabb + 6 * (size_t)(T - 1);
    int n = T; float* dst = b->lvl;                                                                          // the pyramid of the reference refit is done with
    while (n > 64 && Lv.n < 5) {
        const int nd = (n + 63) >> 6;
        k_refit_level<<<grid_for(nd, blk), blk, 0, s>>>(n, Lv.a[Lv.n - 1], dst);
        Lv.a[Lv.n++] = dst; dst += 6 * (size_t)nd; n = nd;
    }
    k_refit_ranges<<<grd, blk, 0, s>>>(T, b->p_range, Lv, b->p_aabb);
    MR_LAUNCH_CHECK("bvh_private_step1");
    return 0;
}
static int private_sah_top(mirres_bvh* b, int T, hipStream_t s) {                        // its upper levels rebuilt by binned SAH over the prefix clusters
    const int blk = 256;
    SahState* st = reinterpret_cast<SahState*>(b->sah_state); SahNode* sn = reinterpret_cast<SahNode*>(b->sah_nodes); SahBin* sb = reinterpret_cast<SahBin*>(b->sah_bins);
    k_sah_reset<<<grid_for(2 * MR_SAH_MAXC, blk), blk, 0, s>>>(st, sn, 2 * MR_SAH_MAXC);
    k_sah_clusters<<<256, blk, 0, s>>>(T, b->p_key64, b->p_range, b->p_parent, b->p_aabb, st, sn, b->sah_iref, b->sah_inode, b->sah_top);
    for (int level = 0; level < MR_SAH_TAIL; level++) {
        const bool lds = level < MR_SAH_LDS_LEVELS;                      // <= 2^level nodes in the level
        const int nodes_max = level < 16 ? (1 << level) : MR_SAH_MAXC;                  // <= 2^level nodes, never more than clusters
        if (lds) k_sah_bin<true><<<64, blk, 0, s>>>(level, st, sn, b->sah_iref, b->sah_inode, b->p_aabb, sb);
        else k_sah_bin<false><<<256, blk, 0, s>>>(level, st, sn, b->sah_iref, b->sah_inode, b->p_aabb, sb);
        k_sah_split<<<grid_for(nodes_max < MR_SAH_MAXC ? nodes_max : MR_SAH_MAXC, blk), blk, 0, s>>>(level, st, sn, sb);
        if (lds) k_sah_assign<true><<<64, blk, 0, s>>>(level, st, sn, b->sah_iref, b->sah_inode, b->p_aabb);
        else k_sah_assign<false><<<256, blk, 0, s>>>(level, st, sn, b->sah_iref, b->sah_inode, b->p_aabb);
    }
    k_sah_tail<<<1, 1024, 0, s>>>(MR_SAH_TAIL, st, sn, b->sah_iref, b->sah_inode, b->p_aabb, sb);
    k_sah_install<<<256, blk, 0, s>>>(st, sn, b->sah_top, b->p_info, b->p_aabb);
    MR_LAUNCH_CHECK("bvh_private_sah_top");
    return 0;
}

extern "C" {

const char* mirres_version(void) { return "mirres-mi355x 0.1 (gfx950)"; }
const char* mirres_last_error(void) { return mr::g_err; }

void mirres_default_config(mirres_config_t* c) {
    c->light_tile_count = 128; c->light_tile_size = 1024; c->screen_tile_size = 8; c->initial_light_samples = 32;
    c->initial_brdf_samples = 1; c->max_history = 20; c->neighbor_offset_count = 8192; c->neighbor_count = 5;
    c->gather_radius = 30.f; c->max_bounce = 2; c->vis_near = 0.01f;
}

int mirres_bvh_create(mirres_bvh_t** out, int max_tris) {
    if (!out || max_tris < 2) { set_error("mirres_bvh_create: need max_tris >= 2"); return MIRRES_E_ARG; }
    mirres_bvh* b = new mirres_bvh();
    b->max_tris = max_tris;
    size_t T = (size_t)max_tris;
    MR_HIP(hipMalloc(&b->ele_aabb, sizeof(float) * 6 * T));
    MR_HIP(hipMalloc(&b->extent, sizeof(uint32_t) * 8));
    MR_HIP(hipMalloc(&b->keys_in, sizeof(uint32_t) * T)); MR_HIP(hipMalloc(&b->keys_out, sizeof(uint32_t) * T));
    MR_HIP(hipMalloc(&b->vals_in, sizeof(uint32_t) * T)); MR_HIP(hipMalloc(&b->vals_out, sizeof(uint32_t) * T));
    MR_HIP(hipMalloc(&b->parent, sizeof(int32_t) * (2 * T)));
    MR_HIP(hipMalloc(&b->flags, sizeof(uint32_t) * 2 * (size_t)T));
    MR_HIP(hipMalloc(&b->lvl, sizeof(float) * 6 * ((size_t)T / 63 + 256)));
    MR_HIP(hipMalloc(&b->own_info, sizeof(int32_t) * 3 * (2 * T)));
    MR_HIP(hipMalloc(&b->own_aabb, sizeof(float) * 6 * (2 * T)));
    MR_HIP(hipMalloc(&b->nodes, sizeof(WideNode) * T));
    MR_HIP(hipMalloc(&b->tris, sizeof(TriRec) * T));
    MR_HIP(hipMalloc(&b->nodes4q, sizeof(Node4q) * T));
    MR_HIP(hipMalloc(&b->leaves, sizeof(LeafRec) * ((size_t)T + 1)));      // + the null leaf
    MR_HIP(hipMalloc(&b->p_keys, sizeof(uint32_t) * T)); MR_HIP(hipMalloc(&b->p_vals, sizeof(uint32_t) * T));
    MR_HIP(hipMalloc(&b->p_key64, sizeof(unsigned long long) * T));
    MR_HIP(hipMalloc(&b->p_info, sizeof(int32_t) * 3 * (2 * T))); MR_HIP(hipMalloc(&b->p_aabb, sizeof(float) * 6 * (2 * T)));
    MR_HIP(hipMalloc(&b->p_range, sizeof(uint32_t) * 2 * T));
    MR_HIP(hipMalloc(&b->p_parent, sizeof(int32_t) * 2 * T));
    {   // scratch of the SAH top (k_sah_*): clusters, rebuilt nodes, one level's bins
        const size_t C = MR_SAH_MAXC;
        MR_HIP(hipMalloc(&b->sah_state, sizeof(SahState)));
        MR_HIP(hipMalloc(&b->sah_nodes, sizeof(SahNode) * 2 * C));
        MR_HIP(hipMalloc(&b->sah_bins, sizeof(SahBin) * C * 3 * MR_SAH_BINS));
        MR_HIP(hipMalloc(&b->sah_iref, sizeof(int32_t) * C)); MR_HIP(hipMalloc(&b->sah_inode, sizeof(int32_t) * C)); MR_HIP(hipMalloc(&b->sah_top, sizeof(int32_t) * C));
        std::vector<SahBin> z(C * 3 * MR_SAH_BINS);
        for (auto& e : z) { e.count = 0; for (int q = 0; q < 3; q++) { e.box[q] = 0xffffffffu; e.box[3 + q] = 0u; } }
        MR_HIP(hipMemcpy(b->sah_bins, z.data(), sizeof(SahBin) * z.size(), hipMemcpyHostToDevice));      // every split leaves its bins clean again
    }
    MR_HIP(hipMalloc(&b->top85q, sizeof(Node4q) * 85));
    MR_HIP(hipMalloc(&b->top341q, sizeof(Node4q) * 341));
    MR_HIP(hipMalloc(&b->root_box, sizeof(float) * 8));
    MR_HIP(hipMalloc(&b->work, sizeof(uint32_t) * MR_WSETS * MR_WSET));
    MR_HIP(hipHostMalloc((void**)&b->err, 64, hipHostMallocMapped)); *b->err = 0u;
    b->sort_tmp_bytes = sizeof(uint32_t) * 256 * ((T + MR_RS_TILE - 1) / MR_RS_TILE + 1);     // (digit, tile) counters of the radix sort + the digit totals
    MR_HIP(hipMalloc(&b->sort_tmp, b->sort_tmp_bytes));
    *out = b;
    return MIRRES_OK;
}

void mirres_bvh_destroy(mirres_bvh_t* b) {
    if (!b) return;
    void* ptrs[] = {b->ele_aabb, b->extent, b->keys_in, b->keys_out, b->vals_in, b->vals_out, b->parent, b->flags, b->own_info,
                    b->own_aabb, b->nodes, b->tris, b->root_box, b->sort_tmp, b->work, b->redo[0], b->redo[1], b->dump_pool, b->lvl, b->nodes4q, b->leaves, b->top85q, b->top341q,
                    b->p_keys, b->p_vals, b->p_key64, b->p_info, b->p_aabb, b->p_range, b->p_parent, b->sah_state, b->sah_nodes, b->sah_bins, b->sah_iref, b->sah_inode, b->sah_top};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (b->err) (void)hipHostFree(b->err);
    delete b;
}

int mirres_bvh_build_level(mirres_bvh_t* b, const float* vert, int V, const int32_t* tri, int T, int32_t* info, float* aabb,
                           int32_t* sorted_codes, int private_level, void* stream) {
    if (!b || !vert || !tri) { set_error("mirres_bvh_build: null argument"); return MIRRES_E_ARG; }
    if (T < 2 || T > b->max_tris) { set_error("mirres_bvh_build: T=%d outside [2,%d]", T, b->max_tris); return MIRRES_E_ARG; }
    hipStream_t s = (hipStream_t)stream;
    if (!info) info = b->own_info;
    if (!aabb) aabb = b->own_aabb;
    b->T = T; b->V = V;
    const int blk = 256, grd = grid_for(T, blk);
    k_init_extent<<<1, 64, 0, s>>>(b->extent);
    k_elements<<<(grd < 256 ? grd : 256), blk, 0, s>>>(vert, tri, T, b->ele_aabb, b->extent);
    k_morton<<<grd, blk, 0, s>>>(b->ele_aabb, b->extent, T, b->keys_out, b->vals_out);
    radix_sort_pairs_u32(b->keys_out, b->vals_out, b->keys_in, b->vals_in, (uint32_t*)b->sort_tmp, T, s);
    k_hierarchy<<<grd, blk, 0, s>>>(T, b->keys_out, b->vals_out, b->ele_aabb, info, aabb, b->parent, b->flags, sorted_codes);
    {
        RefitLevels Lv; Lv.n = 1; Lv.a[0] = aabb + 6 * (size_t)(T - 1);
        int n = T; float* dst = b->lvl;
        while (n > 64 && Lv.n < 5) {
            const int nd = (n + 63) >> 6;
            k_refit_level<<<grid_for(nd, blk), blk, 0, s>>>(n, Lv.a[Lv.n - 1], dst);
            Lv.a[Lv.n++] = dst; dst += 6 * (size_t)nd; n = nd;
        }
        k_refit_ranges<<<grd, blk, 0, s>>>(T, b->flags, Lv, aabb);
    }
    k_pack<<<grd, blk, 0, s>>>(T, info, aabb, vert, tri, b->nodes, b->tris, b->root_box);
    // the 4-wide layout of the shadow-ray / ordered closest-hit kernels: collapsed from the private hierarchy — extended-Morton tree with its upper levels rebuilt by
    // the binned-SAH top (2, default), the plain extended-Morton tree (MIRRES_PRIVATE_TREE=1) — or (0) from the reference LBVH itself
    static const int private_tree_env = [] { const char* e = getenv("MIRRES_PRIVATE_TREE"); return e ? atoi(e) : 2; }();
    int level = private_level >= 0 ? private_level : private_tree_env;
    if (level > 2) level = 2;
    if (T < 8) level = 0;
    if (level >= 1) {
        int rc = private_step1(b, T, aabb, s); if (rc) return rc;
        if (level == 2) { rc = private_sah_top(b, T, s); if (rc) return rc; }
        k_pack4q<true><<<grd, blk, 0, s>>>(T, b->p_info, b->p_aabb, info, aabb, vert, tri, b->nodes4q, b->leaves);
    } else k_pack4q<false><<<grd, blk, 0, s>>>(T, info, aabb, info, aabb, vert, tri, b->nodes4q, b->leaves);
    b->private_level = level;
    if (T - 1 >= 341 * 4) { k_top4q<<<1, 256, 0, s>>>(T, b->nodes4q, b->top85q, 85); k_top4q<<<1, 256, 0, s>>>(T, b->nodes4q, b->top341q, 341); }
    MR_LAUNCH_CHECK("bvh_build");
    return MIRRES_OK;
}

int mirres_bvh_build(mirres_bvh_t* b, const float* vert, int V, const int32_t* tri, int T, int32_t* info, float* aabb,
                     int32_t* sorted_codes, void* stream) {
    return mirres_bvh_build_level(b, vert, V, tri, T, info, aabb, sorted_codes, -1, stream);
}

int mirres_bvh_private_level(mirres_bvh_t* b) { return b ? b->private_level : -1; }

// step 2 on top of a level-1 build (same arrays as the build call): binned-SAH top + the 4-wide collapse again. Every hierarchy gives the same answers.
int mirres_bvh_upgrade(mirres_bvh_t* b, const float* vert, const int32_t* tri, const int32_t* info, const float* aabb, void* stream) {
    if (!b || !vert || !tri) { set_error("mirres_bvh_upgrade: null argument"); return MIRRES_E_ARG; }
    if (b->T < 2) { set_error("mirres_bvh_upgrade: BVH not built"); return MIRRES_E_STATE; }
    if (b->private_level != 1) return MIRRES_OK;                  // nothing to add (level 2 already, or a level-0 build / tiny mesh that has no private tree)
    hipStream_t s = (hipStream_t)stream;
    if (!info) info = b->own_info;
    if (!aabb) aabb = b->own_aabb;
    const int T = b->T, blk = 256, grd = grid_for(T, blk);
    int rc = private_sah_top(b, T, s); if (rc) return rc;
    k_pack4q<true><<<grd, blk, 0, s>>>(T, b->p_info, b->p_aabb, info, aabb, vert, tri, b->nodes4q, b->leaves);
    if (T - 1 >= 341 * 4) { k_top4q<<<1, 256, 0, s>>>(T, b->nodes4q, b->top85q, 85); k_top4q<<<1, 256, 0, s>>>(T, b->nodes4q, b->top341q, 341); }
    b->private_level = 2;
    MR_LAUNCH_CHECK("bvh_upgrade");
    return MIRRES_OK;
}

// development aid (not part of include/mirres.h): counters of the last SAH-top build: clusters, nodes above the cut, rebuilt nodes, internal ones, resolved clusters, failure flag, levels used
int mirres_debug_sah_state(mirres_bvh_t* b, uint32_t* h_out) {
    if (!b || !h_out || !b->sah_state) return MIRRES_E_ARG;
    SahState st;
    MR_HIP(hipDeviceSynchronize());
    MR_HIP(hipMemcpy(&st, b->sah_state, sizeof(st), hipMemcpyDeviceToHost));
    h_out[0] = st.n_items; h_out[1] = st.n_top; h_out[2] = st.n_nodes; h_out[3] = st.n_internal; h_out[4] = st.n_resolved; h_out[5] = st.fail;
    int lv = 0; while (lv + 1 < MR_SAH_LEVELS + 2 && st.first[lv + 1] > st.first[lv]) lv++;
    h_out[6] = (uint32_t)lv; h_out[7] = 0;
    return MIRRES_OK;
}

// development aid (not part of include/mirres.h): the traversal layout as the kernels read it — nodes4q [T-1], leaves [T+1] (the null leaf last), top85q [85], 64 bytes per
// record — for the structural check of tests/test_gpu_layout.py (every leaf reachable exactly once, every decoded child box around its subtree's exact leaf boxes)
int mirres_debug_layout(mirres_bvh_t* b, void* h_nodes4q, void* h_leaves, void* h_top85q) {
    if (!b || b->T < 2 || !h_nodes4q || !h_leaves) return MIRRES_E_ARG;
    MR_HIP(hipDeviceSynchronize());
    MR_HIP(hipMemcpy(h_nodes4q, b->nodes4q, sizeof(Node4q) * (size_t)(b->T - 1), hipMemcpyDeviceToHost));
    MR_HIP(hipMemcpy(h_leaves, b->leaves, sizeof(LeafRec) * (size_t)(b->T + 1), hipMemcpyDeviceToHost));
    if (h_top85q) MR_HIP(hipMemcpy(h_top85q, b->top85q, sizeof(Node4q) * 85, hipMemcpyDeviceToHost));
    return MIRRES_OK;
}

}  // extern "C"
