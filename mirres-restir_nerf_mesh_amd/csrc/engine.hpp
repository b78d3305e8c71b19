// engine.hpp — internal (non-ABI) declarations shared by the .hip translation units of libmirres.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "../../include/mirres.h"

// persistent-traversal work queue: MR_NQ sub-queue heads, one per 128-byte line (bvh_trace.hip grab_chunk); mirres_bvh::work holds MR_WSETS such sets
#define MR_NQ 32
#define MR_QSTRIDE 32
#define MR_WSET ((MR_NQ + 1) * MR_QSTRIDE)   // + one more line: word MR_NQ * MR_QSTRIDE = bit mask of the sub-queues found empty (grab_chunk)
#define MR_WSETS 19
// Binned-SAH top of the private steering hierarchy (bvh_build.hip): clusters = maximal subtrees whose leaves share MR_SAH_PREFIX key bits (of 38), at most
// MR_SAH_LEVELS levels rebuilt above them. The shadow-ray kernel's private stack (bvh_trace.hip MR_ANY_STACK) is sized from these two.
#define MR_SAH_PREFIX 24
#define MR_SAH_LEVELS 40

namespace mr {

// ---- traversal layout (DESIGN.md §BVH layout). One 64-byte record per INTERNAL node holding both children's
// boxes, so a node visit is a single 64 B fetch; leaves are not nodes: a negative child reference ~slot points
// into the triangle table (slot = position in Morton order).
struct __attribute__((aligned(64))) WideNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;  // >=0 internal node index, <0 : ~leaf_slot
    int32_t pad[2];
};
// Compressed 4-wide node (64 B = half a line, 4 dwordx4 loads instead of 8): child boxes as 8-bit offsets from the node's own min corner in
// power-of-two steps, rounded OUTWARD against the exact decode expression fmaf(q, 2^e, origin) the kernel evaluates, so every decoded box
// contains the LBVH box it stands for. Only used by the shadow-ray kernel, whose result depends on the leaves' own boxes alone
// (bvh_trace.hip): leaves that pass the conservative test are re-tested against their exact box, stored with the triangle in LeafRec.
struct __attribute__((aligned(64))) Node4q {
    float org[3]; float step_x;           // quantisation steps (powers of two) as floats: no unpacking in the traversal (an `and` + a shift per axis and
    uint32_t qlo[3], qhi[3];              // visit otherwise); byte k of qlo[a] / qhi[a] = child k's min / max along axis a
    float step_y, step_z;
    int32_t ref[4];                       // >=0 node id, <0 ~leaf slot; an unused entry refers to the null leaf ~T (leaves[T]: an inverted box no ray passes); inside an LDS-staged prefix: MR_TOPBIT | index
};
struct __attribute__((aligned(64))) LeafRec {  // 64 B: triangle (v0, e1, e2) + the leaf's exact LBVH box + primitive id
    float v0[3], e1[3], e2[3];
    float lo[3], hi[3];
    int32_t prim;
};
struct __attribute__((aligned(16))) TriRec {  // 48 B
    float v0[3], e1[3], e2[3];
    int32_t prim;
    int32_t pad[2];
};
struct __attribute__((aligned(16))) Ray {  // 32 B  (ox,oy,oz,t_min, dx,dy,dz,t_max) — same as the ABI's rays[n,8]
    float ox, oy, oz, tmin, dx, dy, dz, tmax;
};
struct __attribute__((aligned(16))) HitRec {  // 32 B closest-hit record
    float px, py, pz; int32_t hit;
    float nx, ny, nz; float t;
};

struct BvhView {
    const WideNode* nodes; const TriRec* tris; const float* root_box;  // root_box -> aabb[0..5] of node 0
    const Node4q* nodes4q; const LeafRec* leaves; const Node4q* top85q; const Node4q* top341q;   // compressed shadow-ray layout
    int T;
    unsigned long long* dbg;   // optional [2 * waves]: wall-clock start / end of every traversal wave (mirres_debug_wave_times)
    uint32_t* err;             // host-mapped sticky word: set by a traversal kernel that had to drop a deferred subtree (provably never; mirres_render / mirres_bvh_trace refuse to go on once it is set)
};

// Shadow rays given as pixel pairs instead of 32-byte rays (the spatial pass of mirres_render): queue entry j = (pixel a, pixel b) stands for ray 2j — from a's
// position towards b's light sample — and ray 2j + 1 — from b's position towards a's. The traversal kernel forms the ray when a lane takes it from the queue:
// origin = pos + vis_near * dir, dir = oct_decode(light sample), the expressions the generating kernel would have used (put_ray), so the traced ray has the
// same bits and the queue carries 4 bytes per ray instead of 32.
// skip_dead (round 4): a ray whose LIGHT reservoir carries luminance 0 (the emptied reservoir of an occluded initial candidate; lum travels with the sample in the
// packed record) is not traced. Its answer cannot reach the frame: the spatial merge multiplies it into target(lum = 0, .) = fmaxf(0, 0 * brdf) = 0, for any brdf
// value incl. inf / NaN (k_spatial_resolve: candAtOther *= canonicalVis, canonAtOther *= candidateVis), and reads it nowhere else (vcode is taken from it only for a
// SELECTED sample, which needs w > 0, i.e. lum > 0). The reference traces these rays and throws the answers away.
// (The further rule "direction in or below the horizon of the origin pixel's shading normal" — eval_brdf is exactly 0 there — was measured too: 1.4 % / 0.3 % more
// rays on the two bench meshes, nothing on the frame, one more 16-byte gather per ray: not kept. Likewise "the origin pixel holds the same light sample and
// already knows the answer (vcode)": 0.2 % of the rays. DESIGN.md Appendix A.)
struct RaySrc { const float4* grec; const float4* rrec; float vis_near; int skip_dead; };   // grec: 64-B pixel records (pos in the fourth quarter), rrec: 32-B packed reservoirs {light_data.xyz, lum | M, weight, vcode, inv_pdf}

}  // namespace mr

struct mirres_bvh {
    int max_tris = 0, T = 0, V = 0;
    int private_level = 0;          // what the traversal layout was collapsed from: 0 reference LBVH, 1 extended-Morton tree, 2 + binned-SAH top (mirres_bvh_upgrade)
    // build workspace
    float* ele_aabb = nullptr;      // [T,6]
    uint32_t* extent = nullptr;     // [6] order-preserving uint encoding of min xyz / max xyz
    uint32_t *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr, *vals_out = nullptr;  // [T] Morton codes / element ids: *_out = k_morton output and, after the sort, the sorted pairs; *_in = the other half of the ping-pong
    int32_t* parent = nullptr;      // [2T-1]
    uint32_t* flags = nullptr;      // [2T] (first, last) sorted-leaf range of every internal node (k_hierarchy -> k_refit_ranges)
    float* lvl = nullptr;           // 64-ary union pyramid over the sorted leaf boxes (k_refit_level)
    void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0;   // radix sort: (digit, tile) counters
    int32_t* own_info = nullptr; float* own_aabb = nullptr;  // used when the caller passes NULL
    // traversal layout
    mr::WideNode* nodes = nullptr;  // [T-1]
    mr::TriRec* tris = nullptr;     // [T]
    mr::Node4q* nodes4q = nullptr;  // [T-1] compressed 4-wide nodes (shadow rays)
    mr::LeafRec* leaves = nullptr;  // [T]
    mr::Node4q* top85q = nullptr, *top341q = nullptr;   // [85], [341] breadth-first prefixes of nodes4q, children inside tagged MR_TOPBIT
    // private steering hierarchy (bvh_build.hip k_emc_*): extended-Morton keys / slots, 64-bit keys, node arrays in the reference's own layout (leaf info[.][2] = slot)
    uint32_t *p_keys = nullptr, *p_vals = nullptr, *p_range = nullptr; unsigned long long* p_key64 = nullptr; int32_t* p_info = nullptr; float* p_aabb = nullptr;
    int32_t* p_parent = nullptr; void *sah_state = nullptr, *sah_nodes = nullptr, *sah_bins = nullptr; int32_t *sah_iref = nullptr, *sah_inode = nullptr, *sah_top = nullptr;   // SAH top over prefix clusters (k_sah_*)
    float* root_box = nullptr;      // [6]
    uint32_t* work = nullptr;       // [MR_WSETS * MR_WSET] head sets of the persistent traversal kernels: 0/1 chain, 2/3 API, 4-6 ordered closest + redo, 7/8 bulk stream, 9/10 path-tracing stream,
                                    // 11 final-stage stream, 12-14 ordered closest + redo and 15/16 any / closest of the second path-tracing stream, 17/18 the second / third
                                    // chain stream of the band pipeline (render.hip)
    unsigned long long* dbg = nullptr;   // see BvhView::dbg
    uint32_t* err = nullptr;             // see BvhView::err (hipHostMalloc, mapped: the host reads it without a copy)
    char* dump_pool = nullptr; size_t dump_pool_bytes = 0;   // mirres_dump_render: shadow rays / results / slots of one pixel chunk
    uint32_t* redo[2] = {nullptr, nullptr}; size_t redo_cap[2] = {0, 0};   // ray ids handed back by the ordered closest-hit fast path (one list per path-tracing stream)
    mr::BvhView view() const { mr::BvhView v; v.nodes = nodes; v.tris = tris; v.root_box = root_box; v.T = T; v.nodes4q = nodes4q; v.leaves = leaves; v.top85q = top85q; v.top341q = top341q; v.dbg = dbg; v.err = err; return v; }
};

// One unit of the band pipeline (round 6, render.hip): the spatial pass of one sample restricted to a band of rows, on queue resources of its own so that units of
// consecutive samples can be in flight on different streams. set 0 = the context's own buffers and head set 0.
struct ChainSet {
    mr::Ray* q = nullptr; int32_t* hit = nullptr; uint32_t* counter = nullptr; int32_t* slot = nullptr; uint32_t* mask = nullptr;
    int head_set = 0; bool clean = false;      // clean: the set's last resolve left the ray counter and the work heads zeroed
};
struct SpatialBand { int y0, y1, gen_y1; ChainSet* set; };   // the resolve covers rows [y0, y1), the generator [y0, gen_y1) (one more row: the fused temporal merge of the
                                                             // band's last row may recompute the spatial merge of the pixel below, whose rays must be in THIS unit's queue)

struct mirres_ctx {
    int fx = 0, fy = 0; size_t N = 0;
    mirres_config_t cfg;
    // wavefront queues
    mr::Ray* any_rays = nullptr;  size_t any_cap = 0;   // shadow rays (<= 10 per pixel in the spatial pass)
    int32_t* any_hit = nullptr;
    mr::Ray* cl_rays = nullptr;   size_t cl_cap = 0;    // closest-hit rays (<= 1 per pixel)
    mr::HitRec* cl_hit = nullptr;
    uint32_t* counters = nullptr;   // [8] device: any_count, closest_count, scratch...
    unsigned long long* stats = nullptr;  // [8] device totals (see mirres_ctx_stats)
    int instrument = 0;
    std::vector<hipEvent_t> ev_any, ev_cl;   // start/stop pairs of event-timed traversal launches
    size_t ev_any_used = 0, ev_cl_used = 0;
    // per-pixel scratch
    int32_t* slot_a = nullptr;      // [N] first any-ray slot of the pixel (or -1)
    uint32_t* mask_a = nullptr;     // [N] per-pass bit mask (spatial: accepted neighbours; bounce: ray kinds)
    int32_t* slot_c = nullptr;      // [N] closest-ray slot (or -1)
    float* pend = nullptr;          // [N,18] pending NEE / BSDF contributions of the bounce pass
    float* noff = nullptr;          // [neighbor_offset_count,2]
    float* tile_aux = nullptr;      // [tiles,8] per tile sample: {light direction xyz, luminance of its radiance | pdf, light_data xyz}
    // frame buffers of the fused loop (mirres_render)
    float* pool = nullptr; size_t pool_floats = 0;
    // K-sample batch of the path-tracing stages (mirres_render): queues + per-slot state for K * N sample slots
    char* ptb = nullptr; size_t ptb_bytes = 0; int ptb_kcap = 0, ptb_kcap_age = 0, ptb_retry_wait = 64;   // ptb_kcap: largest batch the device could hold when an allocation last fell back (0 = never); clamped frames between speculative retries
    int y_off = 0, full_fy = 0;     // strip sharding (mirres_render): global row of local row 0 and the global height; full_fy == 0: the frame is the whole image
    int row_a = 0, row_b = 0, row_mode = 0;   // spatial pass restricted to local rows: 0 all, 1 inside [row_a, row_b), 2 outside (mirres_render's strip_overlap)
    hipStream_t halo_stream = nullptr; hipEvent_t ev_halo[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_halo_t; size_t ev_halo_t_used = 0;   // event pairs around the sampled native halo exchanges of the last frame (mirres_ctx_halo_time)
    const float* occ_own = nullptr; // strip sharding: occupancy with the halo rows zeroed (own-pixel tests of the spatial pass); NULL otherwise
    const float* grec = nullptr;    // set by mirres_render for the duration of a frame: packed 64-byte G records for the neighbour gathers of k_spatial_resolve
    bool chain_reset = false, chain_clean = false;   // mirres_render's chain: k_spatial_resolve leaves the ray counter and the lane-0 work heads zeroed for the next sample
    hipStream_t aux_stream = nullptr, pt_stream = nullptr, pt_stream2 = nullptr, fin_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join_pt = nullptr, ev_join_pt2 = nullptr, ev_join_fin = nullptr;
    std::vector<hipEvent_t> ev_pt;   // k_pt_reduce hand-over between the two path-tracing streams
    std::vector<hipEvent_t> ev_sync; // cross-stream hand-offs of the batch pipeline (mirres_render)
    // band pipeline of the per-sample chain (render.hip): up to two more chain streams with their own queue sets, one event per unit of a batch
    ChainSet chain_sets[3]; void* chain_mem[2] = {nullptr, nullptr}; hipStream_t chain_streams[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_band;
};

namespace mr {
const char* set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
#define MR_HIP(x) do { int _rc = mr::check_hip((x), #x); if (_rc) return _rc; } while (0)
#define MR_LAUNCH_CHECK(name) MR_HIP(hipGetLastError())
// sticky traversal error (BvhView::err): checked where a caller can still be told
inline int bvh_sticky_error(const mirres_bvh* bvh, const char* who) {
    if (bvh && bvh->err && *(volatile uint32_t*)bvh->err) { set_error("%s: an earlier traversal launch overflowed its private stack (flag %u): results since then are not trustworthy", who, *(volatile uint32_t*)bvh->err); return MIRRES_E_STATE; }
    return MIRRES_OK;
}

// queue tracing (bvh_trace.hip). count is read on the device; capacity bounds the grid-stride loop.
int trace_any_items_queue(const mirres_bvh* bvh, const uint2* items, const RaySrc& src, const uint32_t* d_count, size_t capacity, int32_t* hit,
                          unsigned long long* stats, hipStream_t s, int lane, int timed, bool heads_clean, int head_set = -1);
int trace_any_queue(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit,
                    unsigned long long* stats, hipStream_t s, int lane = 0, int timed = 0, bool heads_clean = false);
int trace_any_front_queue(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit, hipStream_t s);
int trace_closest_queue(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, HitRec* out,
                        unsigned long long* stats, hipStream_t s, int lane = 0);
int trace_any(mirres_ctx* ctx, mirres_bvh* bvh, size_t cap, hipStream_t s);       // ctx->any_rays -> ctx->any_hit
int trace_closest(mirres_ctx* ctx, mirres_bvh* bvh, size_t cap, hipStream_t s);   // ctx->cl_rays  -> ctx->cl_hit
// queues and per-slot scratch of the path-tracing stages: the context's own (one sample per pixel — the stepwise ABI) or a K-sample batch
// (mirres_render): slot v = k * N + pixel holds sample k of the batch, so one launch carries K samples' rays.
// scratch of the position bucket sort in front of the material lookup (matnet.hip): two key buffers and a second list (ping-pong of the two radix passes), digit counters
struct GridSort { uint32_t *keys, *keys2; int32_t* sorted; uint32_t* hist; };     // hist: 256 x 1024 workgroup counters + 256 digit totals
struct PtQueues {
    Ray* any_rays; int32_t* any_hit; Ray* cl_rays; HitRec* cl_hit;
    uint32_t* counters;                 // [0] shadow rays, [1] continuation rays, [2] material-net list
    int32_t* slot_a; uint32_t* mask_a; int32_t* slot_c; float* pend;
    int N, NV;                          // pixels, sample slots (K * N)
    int lane;                           // stream of mirres_render the queue is worked on (0 chain, 1 bulk, 2 / 4 path tracing, 3 final stages): own traversal head sets
    int first_sample_is_zero;           // sample 0 of the frame has one pass fewer before the path-tracing stages (no temporal pass)
    // Live-slot lists of the batched path-tracing stages (mirres_render; NULL = every slot, the stepwise ABI): the slots whose path is still going after a
    // resolve stage — a hit, or a specular miss that picks up the environment at the next vertex — in two ping-pong buffers; counters[3 + b] counts list b.
    // Of a K-sample batch's 82 M slots a few million survive the first indirect vertex; the bounce kernels run over the list instead of over all slots.
    int32_t* live[2]; int live_cur;
    GridSort gs;
};
int launch_initial_batch(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_res_t* res, float* tile_data,
                         float* tile_pdf, float* tile_aux, uint32_t frame0, int K, const PtQueues* q, hipStream_t s);
int launch_final_batch(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_env_t* env, const float* occ, const float* pos, const float* normal, const float* ray_dir,
                       const float* kd, const float* rm, const mirres_res_t* res, int K, const PtQueues* q, float* color, float* diff, float* spec, float* tape, hipStream_t s);
int launch_spatial(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_res_t* res, const mirres_res_t* prev_res,
                   const float* neighbor_offsets, uint32_t frameIndex, hipStream_t s, const mirres_res_t* next_res, uint32_t next_frame, const SpatialBand* band = nullptr);
int comm_exchange_halos(void* comm, float* rec, int fx, int n, const int* peer, const int* s0, const int* s1, const int* r0, const int* r1, hipStream_t s);   // comm.hip
int trace_any_q(mirres_ctx* ctx, mirres_bvh* bvh, const Ray* rays, const uint32_t* count, size_t cap, int32_t* hit, hipStream_t s, int lane = 0);
int trace_closest_q(mirres_ctx* ctx, mirres_bvh* bvh, const Ray* rays, const uint32_t* count, size_t cap, HitRec* out, hipStream_t s, int lane = 0);
inline int grid_for(size_t n, int block) { size_t g = (n + block - 1) / block; return (int)(g < 1 ? 1 : g); }
}  // namespace mr
