// treelab — CPU experiment bench for the PRIVATE shadow-ray hierarchy (development aid, not product, not oracle).
// The any-hit answer is independent of the hierarchy over the leaves (bvh_trace.hip: OR over leaves whose own box passes), so the topology of the
// 4-wide tree is free. This tool builds binary hierarchies over the same triangles by several methods, collapses them to 4-wide nodes (greedy by
// area — what k_pack4q does — or SAH-optimal by dynamic programming), and replays the production kernel's visiting order (nearest passing child first,
// the others deferred; leaf = exact box test then triangle test; early exit) on a ray file, counting 64-byte records fetched, box tests and triangle tests.
//   g++ -O2 -fopenmp -std=c++17 scripts/treelab/treelab.cpp -o /tmp/treelab ; /tmp/treelab mesh.bin rays.bin [lbvh.bin]
// mesh.bin: int32 V, T; float32 verts[V*3]; int32 tris[T*3].   rays.bin: int32 n; float32 rays[n*8] (o, tmin, d, tmax).
// lbvh.bin (optional): the reference LBVH as the oracle built it: int32 info[(2T-1)*3]; float32 aabb[(2T-1)*6].
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <vector>

struct Box { float lo[3], hi[3]; };
static inline Box empty_box() { Box b; for (int a = 0; a < 3; a++) { b.lo[a] = 3e38f; b.hi[a] = -3e38f; } return b; }
static inline Box merge(const Box& a, const Box& b) { Box c; for (int k = 0; k < 3; k++) { c.lo[k] = std::min(a.lo[k], b.lo[k]); c.hi[k] = std::max(a.hi[k], b.hi[k]); } return c; }
static inline float area(const Box& b) { float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2]; return dx * dy + dy * dz + dz * dx; }

// binary tree: internal nodes 0..n_int-1, child >= 0 internal, < 0 ~leaf (leaf = triangle slot)
struct BTree { std::vector<int> l, r; std::vector<Box> box; int root = 0; };
struct Mesh { int V, T; std::vector<float> v; std::vector<int> t; std::vector<Box> tb; std::vector<float> cen; Box scene; };

static void refit(BTree& B, const Mesh& M) {
    int n = (int)B.l.size(); B.box.assign(n, empty_box());
    std::vector<int> order; order.reserve(n); std::vector<int> st = {B.root};
    while (!st.empty()) { int x = st.back(); st.pop_back(); order.push_back(x); if (B.l[x] >= 0) st.push_back(B.l[x]); if (B.r[x] >= 0) st.push_back(B.r[x]); }
    for (int i = (int)order.size() - 1; i >= 0; i--) { int x = order[i]; Box a = B.l[x] >= 0 ?      // (nodes a rebuilt top no longer reaches are not in `order`)
         B.box[B.l[x]] : M.tb[~B.l[x]], b = B.r[x] >= 0 ? B.box[B.r[x]] : M.tb[~B.r[x]]; B.box[x] = merge(a, b); }
}
static double sah_binary(const BTree& B, const Mesh& M) {
    double c = 0, ra = area(B.box[B.root]);
    for (size_t i = 0; i < B.l.size(); i++) { c += 1.0 * area(B.box[i]); for (int ch : {B.l[i], B.r[i]}) if (ch < 0) c += 1.0 * area(M.tb[~ch]); }
    return c / ra;
}

static inline uint64_t expand21(uint64_t v) { v &= 0x1fffff; v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull; v = (v | v << 8) & 0x100f00f00f00f00full; v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull; return v; }

// Karras 2012 over sorted 64-bit keys (ties broken by position)
static BTree karras(const std::vector<uint64_t>& key, const std::vector<int>& leaf_of_pos) {
    int n = (int)key.size(); BTree B; B.l.assign(n - 1, 0); B.r.assign(n - 1, 0);
    auto delta = [&](int i, int j) -> int { if (j < 0 || j >= n) return -1; uint64_t a = key[i], b = key[j]; if (a == b) return 64 + __builtin_clz((unsigned)(i ^ j)); return __builtin_clzll(a ^ b); };
    for (int i = 0; i < n - 1; i++) {
        int d = (delta(i, i + 1) - delta(i, i - 1)) >= 0 ? 1 : -1; int dmin = delta(i, i - d); int lmax = 2; while (delta(i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0; for (int t = lmax / 2; t >= 1; t /= 2) if (delta(i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d; int dn = delta(i, j); int s = 0; int t = l;
        do { t = (t + 1) / 2; if (delta(i, i + (s + t) * d) > dn) s += t; } while (t > 1);
        int g = i + s * d + std::min(d, 0);
        B.l[i] = (std::min(i, j) == g) ? ~leaf_of_pos[g] : g; B.r[i] = (std::max(i, j) == g + 1) ? ~leaf_of_pos[g + 1] : g + 1;
    }
    B.root = 0; return B;
}
static BTree build_lbvh(const Mesh& M, int bits) {
    int T = M.T; std::vector<uint64_t> code(T); std::vector<int> idx(T); std::iota(idx.begin(), idx.end(), 0);
    for (int i = 0; i < T; i++) { uint64_t c[3]; for (int a = 0; a < 3; a++) { float u = (M.cen[3 * i + a] - M.scene.lo[a]) / (M.scene.hi[a] - M.scene.lo[a]); double s = (double)(1u << bits); double q = std::min(std::max((double)u * s, 0.0), s - 1); c[a] = (uint64_t)q; }
        code[i] = expand21(c[0]) << 2 | expand21(c[1]) << 1 | expand21(c[2]); }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return code[a] < code[b]; });
    std::vector<uint64_t> key(T); for (int i = 0; i < T; i++) key[i] = code[idx[i]];
    return karras(key, idx);
}
// extended Morton codes (Vinkler, Bittner, Havran 2017): the triangle's size is a fourth coordinate whose bits are interleaved every `every` position triples, so
// that large triangles split off near the top instead of inflating the boxes of the small ones around them
static std::vector<int> g_emc_order;
static std::vector<uint64_t> g_emc_keys;
static BTree build_emc(const Mesh& M, int every, int sbits, int first, int qmode = 0, int xyzbits = 20) {
    int T = M.T; std::vector<uint64_t> code(T); std::vector<int> idx(T); std::iota(idx.begin(), idx.end(), 0);
    float sd = 0; for (int a = 0; a < 3; a++) { float e = M.scene.hi[a] - M.scene.lo[a]; sd += e * e; } sd = std::sqrt(sd);
    for (int i = 0; i < T; i++) { uint32_t c[3]; for (int a = 0; a < 3; a++) { float u = (M.cen[3 * i + a] - M.scene.lo[a]) / (M.scene.hi[a] - M.scene.lo[a]); c[a] = (uint32_t)std::min(std::max((double)u * 1048576.0, 0.0), 1048575.0); }
        float dg = 0; for (int a = 0; a < 3; a++) { float e = M.tb[i].hi[a] - M.tb[i].lo[a]; dg += e * e; } dg = std::sqrt(dg) / sd;
        double qq = qmode == 0 ? dg : (qmode == 1 ? std::sqrt(dg) : std::max(0.0, 1.0 + std::log2(std::max((double)dg, 1e-9)) / 12.0));
        uint32_t q = (uint32_t)std::min(std::max(qq * (double)(1u << sbits), 0.0), (double)((1u << sbits) - 1));
        uint64_t k = 0; int nb = 0, sb = sbits - 1;
        for (int lvl = 19; lvl >= 20 - xyzbits && nb < 61; lvl--) {
            if (sb >= 0 && (19 - lvl) >= first && ((19 - lvl - first) % every) == 0 && nb < 63) { k = k << 1 | ((q >> sb) & 1); sb--; nb++; }
            for (int a = 0; a < 3 && nb < 63; a++) { k = k << 1 | ((c[a] >> lvl) & 1); nb++; } }
        code[i] = k << (63 - nb); }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return code[a] < code[b]; });
    std::vector<uint64_t> key(T); for (int i = 0; i < T; i++) key[i] = code[idx[i]];
    g_emc_order = idx; g_emc_keys = key;
    return karras(key, idx);
}
static BTree from_reference(const Mesh& M, const std::vector<int>& info) {   // reference LBVH arrays -> BTree (leaf = primitive id)
    int T = M.T; BTree B; B.l.assign(T - 1, 0); B.r.assign(T - 1, 0);
    for (int i = 0; i < T - 1; i++) { int L = info[3 * i], R = info[3 * i + 1]; B.l[i] = L >= T - 1 ? ~info[3 * L + 2] : L; B.r[i] = R >= T - 1 ? ~info[3 * R + 2] : R; }
    B.root = 0; return B;
}
// hybrid (HLBVH with a SAH top): the LBVH subtrees whose leaves share a Morton prefix of `bits` bits become items of a top-down binned SAH build
static BTree build_sah_items(const std::vector<Box>& ib, const std::vector<int>& iref, BTree B, int nbins, int mode = 0);
static BTree build_hybrid_keys(const Mesh& M, const std::vector<uint64_t>& key, const std::vector<int>& idx, int prefix_bits_from_top, int mode, int nbins);
static BTree build_hybrid(const Mesh& M, int prefix_bits, int mode = 0, int nbins = 32) {
    int T = M.T; std::vector<uint64_t> code(T); std::vector<int> idx(T); std::iota(idx.begin(), idx.end(), 0);
    for (int i = 0; i < T; i++) { uint64_t c[3]; for (int a = 0; a < 3; a++) { float u = (M.cen[3 * i + a] - M.scene.lo[a]) / (M.scene.hi[a] - M.scene.lo[a]); c[a] = (uint64_t)std::min(std::max((double)u * 1024.0, 0.0), 1023.0); } code[i] = expand21(c[0]) << 2 | expand21(c[1]) << 1 | expand21(c[2]); }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return code[a] < code[b]; });
    std::vector<uint64_t> key(T); for (int i = 0; i < T; i++) key[i] = code[idx[i]] << 34;      // codes left-aligned: prefix lengths count from bit 63
    return build_hybrid_keys(M, key, idx, prefix_bits, mode, nbins);
}
static BTree build_hybrid_keys(const Mesh& M, const std::vector<uint64_t>& key, const std::vector<int>& idx, int prefix_bits, int mode, int nbins) {
    int T = M.T;
    BTree B = karras(key, idx); refit(B, M);
    // leaf range of every internal node (first, last sorted position) by a post-order pass
    int n = T - 1; std::vector<int> first(n), last(n); std::vector<int> pos_of(T); for (int i = 0; i < T; i++) pos_of[idx[i]] = i;
    std::vector<int> order; std::vector<int> st = {B.root};
    while (!st.empty()) { int x = st.back(); st.pop_back(); order.push_back(x); if (B.l[x] >= 0) st.push_back(B.l[x]); if (B.r[x] >= 0) st.push_back(B.r[x]); }
    for (int k = n - 1; k >= 0; k--) { int x = order[k]; int fl = B.l[x] >= 0 ? first[B.l[x]] : pos_of[~B.l[x]], lr = B.r[x] >= 0 ? last[B.r[x]] : pos_of[~B.r[x]]; first[x] = fl; last[x] = lr; }
    std::vector<Box> ib; std::vector<int> iref; st = {B.root};
    while (!st.empty()) { int x = st.back(); st.pop_back();
        auto emit = [&](int ref) { ib.push_back(ref >= 0 ? B.box[ref] : M.tb[~ref]); iref.push_back(ref); };
        int common = key[first[x]] == key[last[x]] ? 64 : __builtin_clzll(key[first[x]] ^ key[last[x]]);
        if (common >= prefix_bits) { emit(x); continue; }
        for (int ch : {B.l[x], B.r[x]}) { if (ch < 0) emit(ch); else st.push_back(ch); } }
    printf("   hybrid %d bits: %zu clusters\n", prefix_bits, ib.size());
    return build_sah_items(ib, iref, B, nbins, mode);
}
static BTree build_sah_items(const std::vector<Box>& ib, const std::vector<int>& iref, BTree B, int nbins, int mode) {
    // new internal nodes are appended to B (the LBVH, whose subtrees the items refer to); the root moves to the top of the SAH tree
    int n = (int)ib.size(); std::vector<int> idx(n); std::iota(idx.begin(), idx.end(), 0);
    std::vector<float> cen(3 * (size_t)n); for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) cen[3 * i + a] = ib[i].lo[a] + 0.5f * (ib[i].hi[a] - ib[i].lo[a]);
    struct Job { int lo, hi, parent, side; }; std::vector<Job> st = {{0, n, -1, 0}};
    while (!st.empty()) {
        Job j = st.back(); st.pop_back(); int m = j.hi - j.lo;
        auto attach = [&](int ref) { if (j.parent >= 0) (j.side ? B.r : B.l)[j.parent] = ref; else B.root = ref; };
        if (m == 1) { attach(iref[idx[j.lo]]); continue; }
        int me = (int)B.l.size(); B.l.push_back(0); B.r.push_back(0); attach(me);
        Box cb = empty_box(); for (int i = j.lo; i < j.hi; i++) for (int a = 0; a < 3; a++) { float c = cen[3 * idx[i] + a]; cb.lo[a] = std::min(cb.lo[a], c); cb.hi[a] = std::max(cb.hi[a], c); }
        int best_axis = -1, best_bin = -1; double best = 1e300;
        if (mode == 1 && m > 2) {   // spatial median of the centroid bounds on the largest axis
            int a = 0; for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k;
            if (cb.hi[a] - cb.lo[a] > 0) { best_axis = a; best_bin = nbins / 2 - 1; }
        }
        int amax = 0; for (int k2 = 1; k2 < 3; k2++) if (cb.hi[k2] - cb.lo[k2] > cb.hi[amax] - cb.lo[amax]) amax = k2;
        if ((mode == 0 || mode == 3) && m > 2) for (int a = 0; a < 3; a++) {
            if (mode == 3 && a != amax) continue;
            float ext = cb.hi[a] - cb.lo[a]; if (!(ext > 0)) continue;
            std::vector<Box> bb(nbins, empty_box()); std::vector<int> cnt(nbins, 0);
            for (int i = j.lo; i < j.hi; i++) { int b = std::min(nbins - 1, (int)((cen[3 * idx[i] + a] - cb.lo[a]) / ext * nbins)); bb[b] = merge(bb[b], ib[idx[i]]); cnt[b]++; }
            std::vector<double> la(nbins), ra(nbins); std::vector<int> lc(nbins), rc(nbins); Box acc = empty_box(); int c = 0;
            for (int b = 0; b < nbins; b++) { if (cnt[b]) acc = merge(acc, bb[b]); c += cnt[b]; la[b] = c ? area(acc) : 0; lc[b] = c; }
            acc = empty_box(); c = 0; for (int b = nbins - 1; b >= 0; b--) { if (cnt[b]) acc = merge(acc, bb[b]); c += cnt[b]; ra[b] = c ? area(acc) : 0; rc[b] = c; }
            for (int b = 0; b + 1 < nbins; b++) { if (!lc[b] || !rc[b + 1]) continue; double cost = la[b] * lc[b] + ra[b + 1] * rc[b + 1]; if (cost < best) { best = cost; best_axis = a; best_bin = b; } }
        }
        int mid;
        if (best_axis < 0) { int a = 0; for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k; mid = j.lo + m / 2; std::nth_element(idx.begin() + j.lo, idx.begin() + mid, idx.begin() + j.hi, [&](int x, int y) { return cen[3 * x + a] < cen[3 * y + a]; }); }
        else { float ext = cb.hi[best_axis] - cb.lo[best_axis]; mid = (int)(std::partition(idx.begin() + j.lo, idx.begin() + j.hi, [&](int x) { return std::min(nbins - 1, (int)((cen[3 * x + best_axis] - cb.lo[best_axis]) / ext * nbins)) <= best_bin; }) - idx.begin()); if (mid == j.lo || mid == j.hi) mid = j.lo + m / 2; }
        st.push_back({j.lo, mid, me, 0}); st.push_back({mid, j.hi, me, 1});
    }
    return B;
}
// top-down binned SAH, one triangle per leaf
static BTree build_sah(const Mesh& M, int nbins = 32) {
    int T = M.T; BTree B; B.l.reserve(T); B.r.reserve(T); std::vector<int> idx(T); std::iota(idx.begin(), idx.end(), 0);
    struct Job { int lo, hi, parent, side; }; std::vector<Job> st = {{0, T, -1, 0}};
    while (!st.empty()) {
        Job j = st.back(); st.pop_back(); int n = j.hi - j.lo;
        auto attach = [&](int ref) { if (j.parent >= 0) (j.side ? B.r : B.l)[j.parent] = ref; };
        if (n == 1) { attach(~idx[j.lo]); continue; }
        int me = (int)B.l.size(); B.l.push_back(0); B.r.push_back(0); attach(me); if (j.parent < 0) B.root = me;
        Box cb = empty_box(); for (int i = j.lo; i < j.hi; i++) for (int a = 0; a < 3; a++) { float c = M.cen[3 * idx[i] + a]; cb.lo[a] = std::min(cb.lo[a], c); cb.hi[a] = std::max(cb.hi[a], c); }
        int best_axis = -1, best_bin = -1; double best = 1e300;
        if (n > 2) for (int a = 0; a < 3; a++) {
            float ext = cb.hi[a] - cb.lo[a]; if (!(ext > 0)) continue;
            std::vector<Box> bb(nbins, empty_box()); std::vector<int> cnt(nbins, 0);
            for (int i = j.lo; i < j.hi; i++) { int b = std::min(nbins - 1, (int)((M.cen[3 * idx[i] + a] - cb.lo[a]) / ext * nbins)); bb[b] = merge(bb[b], M.tb[idx[i]]); cnt[b]++; }
            std::vector<double> la(nbins), ra(nbins); std::vector<int> lc(nbins), rc(nbins); Box acc = empty_box(); int c = 0;
            for (int b = 0; b < nbins; b++) { if (cnt[b]) acc = merge(acc, bb[b]); c += cnt[b]; la[b] = c ? area(acc) : 0; lc[b] = c; }
            acc = empty_box(); c = 0; for (int b = nbins - 1; b >= 0; b--) { if (cnt[b]) acc = merge(acc, bb[b]); c += cnt[b]; ra[b] = c ? area(acc) : 0; rc[b] = c; }
            for (int b = 0; b + 1 < nbins; b++) { if (!lc[b] || !rc[b + 1]) continue; double cost = la[b] * lc[b] + ra[b + 1] * rc[b + 1]; if (cost < best) { best = cost; best_axis = a; best_bin = b; } }
        }
        int mid;
        if (best_axis < 0) { int a = 0; for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k; mid = j.lo + n / 2; std::nth_element(idx.begin() + j.lo, idx.begin() + mid, idx.begin() + j.hi, [&](int x, int y) { return M.cen[3 * x + a] < M.cen[3 * y + a]; }); }
        else { float ext = cb.hi[best_axis] - cb.lo[best_axis]; mid = (int)(std::partition(idx.begin() + j.lo, idx.begin() + j.hi, [&](int x) { return std::min(nbins - 1, (int)((M.cen[3 * x + best_axis] - cb.lo[best_axis]) / ext * nbins)) <= best_bin; }) - idx.begin()); if (mid == j.lo || mid == j.hi) mid = j.lo + n / 2; }
        st.push_back({j.lo, mid, me, 0}); st.push_back({mid, j.hi, me, 1});
    }
    return B;
}
// PLOC (Meister & Bittner 2018) over Morton order, search radius R
static BTree build_ploc(const Mesh& M, int R = 16, int bits = 21, const std::vector<int>* order = nullptr) {
    int T = M.T; std::vector<uint64_t> code(T); std::vector<int> idx(T); std::iota(idx.begin(), idx.end(), 0);
    for (int i = 0; i < T; i++) { uint64_t c[3]; for (int a = 0; a < 3; a++) { float u = (M.cen[3 * i + a] - M.scene.lo[a]) / (M.scene.hi[a] - M.scene.lo[a]); double s = (double)(1u << bits); c[a] = (uint64_t)std::min(std::max((double)u * s, 0.0), s - 1); } code[i] = expand21(c[0]) << 2 | expand21(c[1]) << 1 | expand21(c[2]); }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return code[a] < code[b]; });
    if (order) idx = *order;
    BTree B; B.l.reserve(T); B.r.reserve(T); std::vector<int> ref(T); std::vector<Box> bx(T); for (int i = 0; i < T; i++) { ref[i] = ~idx[i]; bx[i] = M.tb[idx[i]]; }
    int n = T; std::vector<int> nn(T);
    while (n > 1) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) { float best = 3e38f; int bj = -1; for (int j = std::max(0, i - R); j <= std::min(n - 1, i + R); j++) { if (j == i) continue; float a = area(merge(bx[i], bx[j])); if (a < best) { best = a; bj = j; } } nn[i] = bj; }
        std::vector<int> nref; std::vector<Box> nbx; nref.reserve(n); nbx.reserve(n);
        for (int i = 0; i < n; i++) { int j = nn[i]; if (nn[j] == i) { if (i < j) { int me = (int)B.l.size(); B.l.push_back(ref[i]); B.r.push_back(ref[j]); nref.push_back(me); nbx.push_back(merge(bx[i], bx[j])); } } else { nref.push_back(ref[i]); nbx.push_back(bx[i]); } }
        ref.swap(nref); bx.swap(nbx); n = (int)ref.size();
    }
    B.root = ref[0]; return B;
}

// ---- 4-wide collapse
struct Node4 { int ref[4]; Box box[4]; int n; };   // ref >= 0 node4 index, < 0 ~leaf
struct Tree4 { std::vector<Node4> nodes; int root = 0; };
static Box child_box(const BTree& B, const Mesh& M, int ref) { return ref >= 0 ? B.box[ref] : M.tb[~ref]; }
static Tree4 collapse_greedy(const BTree& B, const Mesh& M) {     // k_pack4q: open the internal entry with the largest area, twice
    Tree4 Q; std::vector<int> map(B.l.size(), -1); std::vector<int> st = {B.root}; std::vector<int> order;
    while (!st.empty()) { int x = st.back(); st.pop_back(); int c[4] = {B.l[x], B.r[x], 0, 0}; int nc = 2;
        for (int round = 0; round < 2; round++) { int pick = -1; float best = -1; for (int k = 0; k < nc; k++) if (c[k] >= 0 && area(B.box[c[k]]) > best) { best = area(B.box[c[k]]); pick = k; } if (pick < 0) break; int o = c[pick]; c[pick] = B.l[o]; c[nc++] = B.r[o]; }
        map[x] = (int)Q.nodes.size(); Node4 q; q.n = nc; for (int k = 0; k < 4; k++) { q.ref[k] = k < nc ? c[k] : 0; if (k < nc) q.box[k] = child_box(B, M, c[k]); } Q.nodes.push_back(q);
        for (int k = 0; k < nc; k++) if (c[k] >= 0) st.push_back(c[k]); }
    for (auto& q : Q.nodes) for (int k = 0; k < q.n; k++) if (q.ref[k] >= 0) q.ref[k] = map[q.ref[k]];
    Q.root = map[B.root]; return Q;
}
// SAH-optimal collapse (dynamic programme over "how many of the parent's slots does this subtree take", Ylitie et al. 2017 for 4 slots)
static Tree4 collapse_sah(const BTree& B, const Mesh& M, double Cn = 1.0, double Ct = 1.0) {
    int n = (int)B.l.size(); std::vector<int> order; std::vector<int> st = {B.root};
    while (!st.empty()) { int x = st.back(); st.pop_back(); order.push_back(x); if (B.l[x] >= 0) st.push_back(B.l[x]); if (B.r[x] >= 0) st.push_back(B.r[x]); }
    std::vector<std::array<double, 5>> t(n); std::vector<std::array<signed char, 5>> choice(n);   // t[x][j]: cheapest way to hang subtree x below a parent using at most j slots; choice: 0 = own node, i>0: split i left / j-i right
    auto T_of = [&](int ref, int j) -> double { if (ref < 0) return Ct * area(M.tb[~ref]); return t[ref][j]; };
    for (int k = n - 1; k >= 0; k--) { int x = order[k]; int L = B.l[x], R = B.r[x];
        double forest[5] = {0, 1e300, 1e300, 1e300, 1e300}; int fsplit[5] = {0, 0, 0, 0, 0};
        for (int j = 2; j <= 4; j++) for (int i = 1; i < j; i++) { double c = T_of(L, i) + T_of(R, j - i); if (c < forest[j]) { forest[j] = c; fsplit[j] = i; } }
        t[x][1] = Cn * area(B.box[x]) + forest[4]; choice[x][1] = 0;
        for (int j = 2; j <= 4; j++) { if (forest[j] < t[x][j - 1]) { t[x][j] = forest[j]; choice[x][j] = (signed char)fsplit[j]; } else { t[x][j] = t[x][j - 1]; choice[x][j] = choice[x][j - 1] == 0 ? 0 : choice[x][j - 1]; if (choice[x][j - 1] != 0) choice[x][j] = -(signed char)(j - 1); } } }
    // extraction: entries(x, j) appends the slot entries of subtree x given j slots
    Tree4 Q; std::vector<int> map(n, -1);
    std::function<void(int, int, std::vector<int>&)> entries = [&](int ref, int j, std::vector<int>& out) {
        if (ref < 0) { out.push_back(ref); return; }
        int jj = j; while (jj > 1 && choice[ref][jj] < 0) jj = -choice[ref][jj];     // "same as with fewer slots"
        if (jj == 1 || choice[ref][jj] == 0) { out.push_back(ref); return; }
        int i = choice[ref][jj]; entries(B.l[ref], i, out); entries(B.r[ref], jj - i, out); };
    std::vector<int> work = {B.root};
    while (!work.empty()) { int x = work.back(); work.pop_back(); if (map[x] >= 0) continue; std::vector<int> e;
        // node x becomes a wide node: its forest with 4 slots
        { int best_i = 1; double best = 1e300; for (int i = 1; i < 4; i++) { double c = T_of(B.l[x], i) + T_of(B.r[x], 4 - i); if (c < best) { best = c; best_i = i; } } entries(B.l[x], best_i, e); entries(B.r[x], 4 - best_i, e); }
        map[x] = (int)Q.nodes.size(); Node4 q; q.n = (int)e.size(); for (int k = 0; k < 4; k++) { q.ref[k] = k < q.n ? e[k] : 0; if (k < q.n) q.box[k] = child_box(B, M, e[k]); } Q.nodes.push_back(q);
        for (int k = 0; k < q.n; k++) if (e[k] >= 0) work.push_back(e[k]); }
    for (auto& q : Q.nodes) for (int k = 0; k < q.n; k++) if (q.ref[k] >= 0) q.ref[k] = map[q.ref[k]];
    Q.root = map[B.root]; return Q;
}
static double sah4(const Tree4& Q, const Mesh& M) { double c = 0; Box rb = empty_box(); const Node4& r = Q.nodes[Q.root]; for (int k = 0; k < r.n; k++) rb = merge(rb, r.box[k]);
    for (auto& q : Q.nodes) { Box b = empty_box(); for (int k = 0; k < q.n; k++) { b = merge(b, q.box[k]); if (q.ref[k] < 0) c += area(q.box[k]); } c += area(b); } return c / area(rb); }

// ---- replay of k_trace_any4q's visiting order
struct Counts { double rec = 0, box = 0, tri = 0, hit = 0, wave_iter = 0; int maxsp = 0; };
static inline bool slab(const Box& b, const float o[3], const float inv[3], float tmin, float tmax, float& tn) {
    float n = tmin, f = tmax; for (int a = 0; a < 3; a++) { float t0 = (b.lo[a] - o[a]) * inv[a], t1 = (b.hi[a] - o[a]) * inv[a]; if (inv[a] < 0) std::swap(t0, t1); n = t0 > n ? t0 : n; f = t1 < f ? t1 : f; } tn = n; return f > n; }
static bool tri_hit(const Mesh& M, int s, const float o[3], const float d[3]) {
    const int* ti = &M.t[3 * s]; const float* v0 = &M.v[3 * ti[0]]; float e1[3], e2[3], P[3], Tv[3], Q[3];
    for (int a = 0; a < 3; a++) { e1[a] = M.v[3 * ti[1] + a] - v0[a]; e2[a] = M.v[3 * ti[2] + a] - v0[a]; Tv[a] = o[a] - v0[a]; }
    P[0] = d[1] * e2[2] - d[2] * e2[1]; P[1] = d[2] * e2[0] - d[0] * e2[2]; P[2] = d[0] * e2[1] - d[1] * e2[0];
    float det = e1[0] * P[0] + e1[1] * P[1] + e1[2] * P[2]; if (det > -1e-15f && det < 1e-15f) return false; float id = 1 / det;
    float u = (Tv[0] * P[0] + Tv[1] * P[1] + Tv[2] * P[2]) * id; if (u < 0 || u > 1) return false;
    Q[0] = Tv[1] * e1[2] - Tv[2] * e1[1]; Q[1] = Tv[2] * e1[0] - Tv[0] * e1[2]; Q[2] = Tv[0] * e1[1] - Tv[1] * e1[0];
    float v = (d[0] * Q[0] + d[1] * Q[1] + d[2] * Q[2]) * id; return !(v < 0 || u + v > 1);
}
static Counts replay(const Tree4& Q, const Mesh& M, const std::vector<float>& rays, std::vector<unsigned char>* hits) {
    int n = (int)rays.size() / 8; std::vector<int> recs(n); Counts C; double rec = 0, box = 0, tri = 0, hit = 0, rec_hit = 0; int maxsp = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : rec, box, tri, hit, rec_hit) reduction(max : maxsp)
    for (int i = 0; i < n; i++) {
        const float* r = &rays[8 * i]; float o[3] = {r[0], r[1], r[2]}, d[3] = {r[4], r[5], r[6]}; float l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); float inv[3];
        for (int a = 0; a < 3; a++) { d[a] /= l; float di = d[a] == 0.f ? 1e-6f : d[a]; inv[a] = 1.0f / di; }
        int stack[256]; int sp = 0; int cur = Q.root; bool have = true, h = false; int nrec = 0;
        while (have) {
            nrec++;
            if (cur < 0) { int s = ~cur; float tn; box += 1; if (slab(M.tb[s], o, inv, r[3], r[7], tn)) { tri += 1; if (tri_hit(M, s, o, d)) { h = true; break; } } if (sp > 0) cur = stack[--sp]; else have = false; continue; }
            const Node4& q = Q.nodes[cur]; int next = 0x7fffffff; float ntn = 0;
            // TL_ORDER: which passing child is visited first. 0 (the kernel's): the nearest, the others deferred in slot order with the nearest-so-far swap;
            // 1 largest surface area; 2 nearest, the others sorted too; 3 longest chord (exit - entry); 9 longest chord, the others in slot order; 11 largest exit distance
            static const int ORDER = getenv("TL_ORDER") ? atoi(getenv("TL_ORDER")) : 0;
            if (ORDER == 0) {
            for (int k = 0; k < q.n; k++) { float tn; box += 1; if (!slab(q.box[k], o, inv, r[3], 3e38f, tn)) continue;
                if (next == 0x7fffffff) { next = q.ref[k]; ntn = tn; } else { int far = q.ref[k]; if (tn < ntn) { far = next; next = q.ref[k]; ntn = tn; } if (sp < 256) stack[sp++] = far; } }
            } else {
                int cr[4]; float ck[4]; int nc = 0;
                for (int k = 0; k < q.n; k++) { float tn; box += 1; if (!slab(q.box[k], o, inv, r[3], 3e38f, tn)) continue;
                    float tf = 3e38f; for (int a = 0; a < 3; a++) { float t0 = (q.box[k].lo[a] - o[a]) * inv[a], t1 = (q.box[k].hi[a] - o[a]) * inv[a]; tf = std::min(tf, std::max(t0, t1)); }
                    float key = tn;
                    if (ORDER == 1) key = -area(q.box[k]);
                    if (ORDER == 3 || ORDER == 9) key = -(tf - tn);
                    if (ORDER == 11) key = -tf;
                    cr[nc] = q.ref[k]; ck[nc] = key; nc++; }
                if (ORDER == 9) { int bi = 0; for (int a = 1; a < nc; a++) if (ck[a] <= ck[bi]) bi = a; if (nc > 0) { next = cr[bi]; for (int a = 0; a < nc; a++) if (a != bi && sp < 256) stack[sp++] = cr[a]; } }
                else { for (int a = 1; a < nc; a++) for (int b = a; b > 0 && ck[b] < ck[b - 1]; b--) { std::swap(ck[b], ck[b - 1]); std::swap(cr[b], cr[b - 1]); }
                       if (nc > 0) { next = cr[0]; for (int a = nc - 1; a >= 1; a--) if (sp < 256) stack[sp++] = cr[a]; } }
            }
            maxsp = std::max(maxsp, sp);
            if (next != 0x7fffffff) cur = next; else if (sp > 0) cur = stack[--sp]; else have = false;
        }
        recs[i] = nrec; rec += nrec; hit += h; if (hits) (*hits)[i] = h; if (h) rec_hit += nrec;
    }
    double wi = 0; for (int w = 0; w + 64 <= n; w += 64) { int m = 0; for (int k = 0; k < 64; k++) m = std::max(m, recs[w + k]); wi += m; }
    if (getenv("TL_ORDER")) printf("      records per occluded ray %.2f, per free ray %.2f\n", rec_hit / std::max(1.0, hit), (rec - rec_hit) / std::max(1.0, n - hit)); C.rec = rec / n; C.box = box / n; C.tri = tri / n; C.hit = hit / n; C.wave_iter = wi / (n / 64); C.maxsp = maxsp; return C;
}
// child boxes as the GPU stores them (Node4q): offsets from the node's corner in power-of-two steps, `bits` bits per plane, rounded outward
static Tree4 quantise(Tree4 Q, int bits) {
    const float qmax = (float)((1 << bits) - 1);
    for (auto& q : Q.nodes) for (int a = 0; a < 3; a++) {
        float lo = 3e38f, hi = -3e38f; for (int k = 0; k < q.n; k++) { lo = std::min(lo, q.box[k].lo[a]); hi = std::max(hi, q.box[k].hi[a]); }
        float r = (hi - lo) / qmax; int e; std::frexp(r > 0 ? r : 1e-30f, &e); float st = std::ldexp(1.0f, e);      // smallest power of two >= r
        while (st * 0.5f >= r && st > 1e-30f) st *= 0.5f; while (lo + st * qmax < hi) st *= 2;
        for (int k = 0; k < q.n; k++) { float ql = std::floor((q.box[k].lo[a] - lo) / st), qh = std::ceil((q.box[k].hi[a] - lo) / st); q.box[k].lo[a] = lo + std::min(std::max(ql, 0.f), qmax) * st; q.box[k].hi[a] = lo + std::min(std::max(qh, 0.f), qmax) * st; }
    }
    return Q;
}
static int depth4(const Tree4& Q) { int d = 0; std::vector<std::pair<int, int>> st = {{Q.root, 1}}; while (!st.empty()) { auto [x, dd] = st.back(); st.pop_back(); d = std::max(d, dd); for (int k = 0; k < Q.nodes[x].n; k++) if (Q.nodes[x].ref[k] >= 0) st.push_back({Q.nodes[x].ref[k], dd + 1}); } return d; }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: treelab mesh.bin rays.bin [lbvh.bin]\n"); return 1; }
    Mesh M; FILE* f = fopen(argv[1], "rb"); if (!f) return 1; if (fread(&M.V, 4, 1, f) != 1 || fread(&M.T, 4, 1, f) != 1) return 1; M.v.resize(3 * (size_t)M.V); M.t.resize(3 * (size_t)M.T);
    if (fread(M.v.data(), 4, M.v.size(), f) != M.v.size() || fread(M.t.data(), 4, M.t.size(), f) != M.t.size()) return 1; fclose(f);
    M.tb.resize(M.T); M.cen.resize(3 * (size_t)M.T); M.scene = empty_box();
    for (int i = 0; i < M.T; i++) { Box b = empty_box(); for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { float x = M.v[3 * M.t[3 * i + k] + a]; b.lo[a] = std::min(b.lo[a], x); b.hi[a] = std::max(b.hi[a], x); } M.tb[i] = b; M.scene = merge(M.scene, b); for (int a = 0; a < 3; a++) M.cen[3 * i + a] = b.lo[a] + 0.5f * (b.hi[a] - b.lo[a]); }
    std::vector<float> rays; { FILE* g = fopen(argv[2], "rb"); int n; if (!g || fread(&n, 4, 1, g) != 1) return 1; rays.resize(8 * (size_t)n); if (fread(rays.data(), 4, rays.size(), g) != rays.size()) return 1; fclose(g); }
    printf("mesh T=%d, rays %zu\n", M.T, rays.size() / 8);
    std::vector<std::pair<std::string, BTree>> trees;
    if (argc > 3) { std::vector<int> info(3 * (size_t)(2 * M.T - 1)); FILE* g = fopen(argv[3], "rb"); if (g && fread(info.data(), 4, info.size(), g) == info.size()) trees.push_back({"reference LBVH (30-bit)", from_reference(M, info)}); if (g) fclose(g); }
    else trees.push_back({"LBVH 30-bit", build_lbvh(M, 10)});
    trees.push_back({"EMC lin s8 (32-bit: xyz8)", build_emc(M, 1, 8, 0, 0, 8)});
    { build_emc(M, 1, 8, 0, 0, 8); std::vector<uint64_t> ek = g_emc_keys; std::vector<int> eo = g_emc_order;
      trees.push_back({"EMC cut 20 bits + SAH8 top", build_hybrid_keys(M, ek, eo, 20, 0, 8)});
      trees.push_back({"EMC cut 24 bits + SAH8 top", build_hybrid_keys(M, ek, eo, 24, 0, 8)});
      trees.push_back({"EMC cut 24 + SAH8 largest axis", build_hybrid_keys(M, ek, eo, 24, 3, 8)});
      trees.push_back({"EMC cut 24 + SAH16 largest axis", build_hybrid_keys(M, ek, eo, 24, 3, 16)});
      trees.push_back({"EMC cut 28 bits + SAH8 top", build_hybrid_keys(M, ek, eo, 28, 0, 8)}); }
    trees.push_back({"hybrid 18: SAH 32 bins", build_hybrid(M, 18, 0, 32)});
    trees.push_back({"hybrid 18: SAH 8 bins", build_hybrid(M, 18, 0, 8)});
    trees.push_back({"hybrid 18: SAH 4 bins", build_hybrid(M, 18, 0, 4)});
    trees.push_back({"hybrid 18: spatial median", build_hybrid(M, 18, 1, 32)});
    trees.push_back({"hybrid 18: object median", build_hybrid(M, 18, 2, 32)});
    trees.push_back({"hybrid 21: SAH 8 bins", build_hybrid(M, 21, 0, 8)});
    trees.push_back({"binned SAH", build_sah(M)});
    if (const char* only = getenv("TL_ONLY")) { std::vector<std::pair<std::string, BTree>> keep; for (auto& t : trees) if (t.first.find(only) != std::string::npos) keep.push_back(t); trees.swap(keep); }
    std::vector<unsigned char> ref_hits(rays.size() / 8), hits(rays.size() / 8);
    printf("%-26s %-8s %9s %9s %7s %8s %8s %8s %9s %6s %6s\n", "binary hierarchy", "collapse", "SAH(bin)", "SAH(4w)", "depth4", "rec/ray", "box/ray", "tri/ray", "wave-it", "maxsp", "hit");
    bool first = true;
    for (auto& [name, B] : trees) { refit(B, M); double sb = sah_binary(B, M);
        for (int c = 0; c < 4; c++) { if (c == 1 && !getenv("TL_DP")) continue; Tree4 Q = c == 1 ? collapse_sah(B, M) : collapse_greedy(B, M); if (c == 2) Q = quantise(Q, 8); if (c == 3) Q = quantise(Q, 12); Counts C = replay(Q, M, rays, first ? &ref_hits : &hits);
            if (!first && hits != ref_hits) printf("!! hit bits differ from the first tree\n"); first = false;
            printf("%-26s %-8s %9.2f %9.2f %7d %8.2f %8.2f %8.2f %9.1f %6d %6.3f\n", name.c_str(), c == 1 ? "sah-dp" : (c == 2 ? "greedy q8" : (c == 3 ? "greedy q12" : "greedy")), sb, sah4(Q, M), depth4(Q), C.rec, C.box, C.tri, C.wave_iter, C.maxsp, C.hit); fflush(stdout); } }
    return 0;
}
