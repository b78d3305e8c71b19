// render.hip — the whole frame without returning to Python: run_restir_di_with_pt + restir_di_with_pt
// (nerf/renderer_restir.py:230-550). The reference issues ~60 launches + 2 torch.where syncs + 9 accumulate kernels per
// spp from Python; here one C call enqueues the spp loop on a stream, accumulations are fused into the producing kernels,
// the material net scatters in place (no index tensors), and nothing synchronises with the host.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "engine.hpp"
#include "device_math.hpp"

namespace mr {

#define MR_BLOCK 256

int launch_final_shading(const mirres_env_t* env, const float* occ, const float* normal, const float* ray_dir, const float* kd, const float* rm,
                         const float* fdir, const float* fdist, const float* fLi, int N, float* color, float* dl, float* sl, bool acc, hipStream_t s);
int launch_new_dir(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_path_t* p, uint32_t frameIndex, uint32_t bounce_count, hipStream_t s, const PtQueues* q);
int launch_bounce(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_env_t* env, const mirres_path_t* p, uint32_t frameIndex, uint32_t bounce_count, float* color,
                  float* dc, float* sc, float* acc_c, float* acc_d, float* acc_s, hipStream_t s, const PtQueues* q);
int launch_direct_bwd(const float* tex, int Wc, int Hc, int N, int S, const float* occ, const float* normal, const float* ray_dir_raw, const float* kd, const float* rm,
                      const float* tape, const float* g_color, const float* g_diff, const float* g_spec, float* g_normal, float* g_kd, float* g_rm, float* g_env,
                      hipStream_t s);
int launch_bilateral5(int fx, int fy, float sigma, const float* const col[5], const float* nrm, const float* zdz, float* scratch, float* const out[5], hipStream_t s);
int launch_eaw5(int fx, int fy, int step, float c_phi, float n_phi, float p_phi, const float* occ, const float* const in[5], const float* normal, const float* pos,
                float* const out[5], hipStream_t s);
int launch_matnet_scatter(const mirres_matnet_t* m, const float* occ, const float* pos, int n, float* kd, float* rm, int use_scale, const float* scale3,
                          const float* const_kd, const float* const_rm, hipStream_t s);
int launch_matnet_scatter_mfma(const mirres_matnet_t* m, const float* occ, const float* pos, int n, float* kd, float* rm, int use_scale, const float* scale3,
                               int32_t* index, uint32_t* count, hipStream_t s, const int32_t* live, const uint32_t* live_count, const GridSort* gs);

// run_restir_di_with_pt :484-486 + restir_di_with_pt :279-287
__global__ void __launch_bounds__(MR_BLOCK) k_prep(int N, float* __restrict__ occ, const float* __restrict__ ray_dir_in, const float* __restrict__ normal,
                                                   const float* __restrict__ depth, const float* __restrict__ kd, const float* __restrict__ rm,
                                                   float* __restrict__ ray_dir, float* __restrict__ nd, float* __restrict__ brdf, const float* __restrict__ pos,
                                                   float4* __restrict__ grec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (occ[i] <= 0.5f) occ[i] = 0.f;
    v3 d = ld3(ray_dir_in, i);
    float l = fmaxf(sqrtf(dot(d, d)), 1e-6f);  // F.normalize(eps=1e-6)
    st3(ray_dir, i, V3(d.x / l, d.y / l, d.z / l));
    nd[4 * (size_t)i] = normal[3 * (size_t)i]; nd[4 * (size_t)i + 1] = normal[3 * (size_t)i + 1]; nd[4 * (size_t)i + 2] = normal[3 * (size_t)i + 2];
    nd[4 * (size_t)i + 3] = depth[i];
    const float m = rm[2 * (size_t)i + 1], r = rm[2 * (size_t)i];
    brdf[3 * (size_t)i] = (kd[3 * (size_t)i] * 0.2126f + kd[3 * (size_t)i + 1] * 0.7152f) + kd[3 * (size_t)i + 2] * 0.0722f;
    brdf[3 * (size_t)i + 1] = (m * 0.2126f + m * 0.7152f) + m * 0.0722f;
    float a = fminf(fmaxf(r, 0.01f), 1.f);
    brdf[3 * (size_t)i + 2] = a * a;
    // the same values as one 64-byte record per pixel for the neighbour gathers of the reuse passes (passes.hip GBufD::rec)
    float4 r0, r1, r2, r3;
    r0.x = nd[4 * (size_t)i]; r0.y = nd[4 * (size_t)i + 1]; r0.z = nd[4 * (size_t)i + 2]; r0.w = nd[4 * (size_t)i + 3];
    r1.x = ray_dir[3 * (size_t)i]; r1.y = ray_dir[3 * (size_t)i + 1]; r1.z = ray_dir[3 * (size_t)i + 2]; r1.w = occ[i];
    r2.x = brdf[3 * (size_t)i]; r2.y = brdf[3 * (size_t)i + 1]; r2.z = brdf[3 * (size_t)i + 2]; r2.w = 0.f;
    r3.x = pos[3 * (size_t)i]; r3.y = pos[3 * (size_t)i + 1]; r3.z = pos[3 * (size_t)i + 2]; r3.w = 0.f;
    grec[4 * (size_t)i] = r0; grec[4 * (size_t)i + 1] = r1; grec[4 * (size_t)i + 2] = r2; grec[4 * (size_t)i + 3] = r3;
}
__global__ void __launch_bounds__(MR_BLOCK) k_flip_env(int W, int H, const float* __restrict__ in, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int y = i / W, x = i % W;
    st3(out, i, ld3(in, (size_t)(H - 1 - y) * W + x));
}
// strip sharding: occupancy with the halo rows zeroed — the stages that only feed this rank's own pixels skip the halo rows like background
__global__ void __launch_bounds__(MR_BLOCK) k_own_occ(int N, int fx, int own_y0, int own_y1, const float* __restrict__ occ, float* __restrict__ occ_own) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int y = i / fx;
    occ_own[i] = (y >= own_y0 && y < own_y1) ? occ[i] : 0.f;
}
// average + combined indirect (:507-515)
__global__ void __launch_bounds__(MR_BLOCK) k_average(size_t n3, float spp, float* t0, float* t1, float* t2, float* t3, float* t4, float* t5, float* comb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    t0[i] /= spp; t1[i] /= spp; t2[i] /= spp; t3[i] /= spp; t4[i] /= spp; t5[i] /= spp;
    comb[i] = t4[i] + t5[i];
}
// final composite (:543-549)
__global__ void __launch_bounds__(MR_BLOCK) k_composite(int N, const float* __restrict__ occ, const float* __restrict__ kd, const float* __restrict__ rm,
                                                        const float* __restrict__ dd, const float* __restrict__ ds, const float* __restrict__ di,
                                                        float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float m = rm[2 * (size_t)i + 1];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float d = kd[3 * (size_t)i + k] * (1.0f - m);
        float v = d * dd[3 * (size_t)i + k] + ds[3 * (size_t)i + k] + di[3 * (size_t)i + k];
        if (occ[i] <= 0.1f) v = 1.0f;
        if (isnan(v)) v = 0.f; else if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;  // torch.nan_to_num
        out[3 * (size_t)i + k] = v;
    }
}

// Adds a batch's indirect light to the frame totals in the order a sample-by-sample loop would: sample k ascending, bounce ascending
// (renderer_restir.py:420-422, 450-452), so the sums are bit-identical to the unbatched loop. cb = [bounce][colour, diffuse, specular][NV * 3].
__global__ void __launch_bounds__(MR_BLOCK) k_pt_reduce(int N, int K, int nb, const float* __restrict__ cb, const uint32_t* __restrict__ maskb, float* __restrict__ t3,
                                                        float* __restrict__ t4, float* __restrict__ t5) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n3 = 3 * (size_t)N;
    if (i >= n3) return;
    const size_t nv3 = n3 * (size_t)K;
    float a = t3[i], b = t4[i], c = t5[i];
    for (int k = 0; k < K; k++)
        for (int bo = 0; bo < nb; bo++) {
            // colours exist only where the bounce had something to add (mask bits 0/1: NEE / BSDF contributions, bit 4: environment pick-up);
            // elsewhere the slot's contribution is zero and x + 0 == x
            if (!(maskb[(size_t)bo * (nv3 / 3) + (size_t)k * N + i / 3] & 19u)) continue;
            const float* base = cb + (size_t)bo * 3 * nv3 + (size_t)k * n3 + i;
            a += base[0]; b += base[nv3]; c += base[2 * nv3];
        }
    t3[i] = a; t4[i] = b; t5[i] = c;
}

// development aid: order-independent checksum of a buffer (MIRRES_DBG_SUM=1 prints one line per ReSTIR stage and sample to stderr)
__global__ void __launch_bounds__(MR_BLOCK) k_checksum(const uint32_t* __restrict__ p, size_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (unsigned long long)p[i] * (unsigned long long)((i % 1021) + 1);
    atomicAdd(out, acc);
}

struct Pool {
    float* base; size_t used, cap;
    float* take(size_t n) { n = (n + 63) & ~(size_t)63; float* p = base + used; used += n; return used <= cap ? p : nullptr; }
};

static size_t pool_need(size_t N, size_t WH, size_t H, size_t TS) {
    size_t per_px = 3 + 4 + 3 + 12 + 1 + 3 + 1 + 3 + 18 + 9 + 5 + 10 + 10 + 5 + 6 + 3 + 16 + 1 + 20;
    return per_px * N + 3 * WH + WH + (WH + H) + H + (H + 1) + 4 * TS + 64 * 64;
}

}  // namespace mr

using namespace mr;

struct FrameBufs {
    float *ray_dir, *nd, *brdf, *grec, *occ_own, *bil;
    float *r_ld[2], *r_pdf[2], *r_w[2]; int32_t* r_M[2];
    float *vis, *fdir, *fdist, *fLi;
    float* tot[6];  // total_color, total_diff, total_spec, total_color_1, total_diff_1, total_spec_1
    float *c1, *d1, *s1;
    float *prd, *new_pos, *new_rd, *new_occ, *new_n, *tmp_pos, *tmp_rd, *tmp_occ, *tmp_n, *new_kd, *new_rm;
    float *tex, *pdf, *cdf, *mpdf, *mcdf, *tile_data, *tile_pdf;
    float *den_a, *den_b, *comb;
};

static int carve(mirres_ctx* ctx, int Wc, int Hc, FrameBufs& B) {
    const size_t N = ctx->N, WH = (size_t)Wc * Hc, TS = (size_t)ctx->cfg.light_tile_count * ctx->cfg.light_tile_size;
    const size_t need = pool_need(N, WH, (size_t)Hc, TS);
    if (ctx->pool_floats < need) {
        if (ctx->pool) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipFree(ctx->pool)); ctx->pool = nullptr; }
        MR_HIP(hipMalloc(&ctx->pool, sizeof(float) * need));
        ctx->pool_floats = need;
    }
    Pool P = {ctx->pool, 0, ctx->pool_floats};
    B.ray_dir = P.take(3 * N); B.nd = P.take(4 * N); B.brdf = P.take(3 * N);
    for (int k = 0; k < 2; k++) { B.r_ld[k] = P.take(3 * N); B.r_pdf[k] = P.take(N); B.r_w[k] = P.take(N); B.r_M[k] = reinterpret_cast<int32_t*>(P.take(N)); }
    B.vis = P.take(N); B.fdir = P.take(3 * N); B.fdist = P.take(N); B.fLi = P.take(3 * N);
    for (int k = 0; k < 6; k++) B.tot[k] = P.take(3 * N);
    B.c1 = P.take(3 * N); B.d1 = P.take(3 * N); B.s1 = P.take(3 * N);
    B.prd = P.take(5 * N); B.new_pos = P.take(3 * N); B.new_rd = P.take(3 * N); B.new_occ = P.take(N); B.new_n = P.take(3 * N);
    B.tmp_pos = P.take(3 * N); B.tmp_rd = P.take(3 * N); B.tmp_occ = P.take(N); B.tmp_n = P.take(3 * N); B.new_kd = P.take(3 * N); B.new_rm = P.take(2 * N);
    B.tex = P.take(3 * WH); B.pdf = P.take(WH); B.cdf = P.take(WH + Hc); B.mpdf = P.take(Hc); B.mcdf = P.take(Hc + 1);
    B.tile_data = P.take(3 * TS); B.tile_pdf = P.take(TS);
    B.den_a = P.take(3 * N); B.den_b = P.take(3 * N); B.comb = B.c1;  // comb reuses c1 after the loop
    B.grec = P.take(16 * N); B.occ_own = P.take(N); B.bil = P.take(20 * N);   // bil: packed taps of the bilateral finish
    if (!B.den_b || !B.grec || !B.occ_own || !B.bil) { set_error("mirres_render: internal pool too small"); return MIRRES_E_STATE; }
    return 0;
}

// K-sample batch of the path-tracing stages: queues + per-slot path state for K * N sample slots, one allocation kept in the context
struct PtBatch {
    int K; PtQueues q;   // path-tracing stages
    PtQueues qf, qv;     // initial / final-visibility stages (each its own rays, results and counter: the stages of different batches may run side by side)
    float *prd, *pos[2], *rd[2], *occ[2], *n[2], *kd, *rm;
    float* cb;      // [max_bounce][3][K * N * 3] per-bounce colour / diffuse / specular of every slot
    uint32_t* maskb; // [max_bounce][K * N] per-bounce ray/colour masks (k_bounce_gen)
    // history-free ReSTIR stages: two sets (batch parity) of K initial / temporal reservoirs and K spatial-output reservoirs, light tiles of K samples
    mirres_res_t rinit[2], rspat[2];
    float *tile_data, *tile_pdf, *tile_aux;
};
static mirres_res_t res_slot(const mirres_res_t& r, int k, size_t N) {   // packed 32-byte records (passes.hip resd(): light_pdf == NULL)
    mirres_res_t o; o.light_data = r.light_data + 8 * (size_t)k * N; o.light_pdf = nullptr; o.M = nullptr; o.weight = nullptr; return o;
}
// The per-slot path-tracing state of a batch can be worked on as two halves (slots [h * cap, (h + 1) * cap) of every array, their own counters):
// two path-tracing streams then each advance a sub-batch of <= cap / N samples
struct PtSet { PtQueues q; float *prd, *pos[2], *rd[2], *occ[2], *n[2], *kd, *rm, *cb; uint32_t* maskb; };
static PtSet pt_set(const PtBatch& PB, int h, size_t cap, int nb) {
    PtSet S; const size_t o = (size_t)h * cap;
    S.q = PB.q;
    S.q.any_rays = PB.q.any_rays + 2 * o; S.q.any_hit = PB.q.any_hit + 2 * o; S.q.cl_rays = PB.q.cl_rays + o; S.q.cl_hit = PB.q.cl_hit + o;
    S.q.counters = PB.q.counters + 8 * h;
    S.q.slot_a = PB.q.slot_a + o; S.q.mask_a = PB.q.mask_a + o; S.q.slot_c = PB.q.slot_c + o; S.q.pend = PB.q.pend + 18 * o;
    S.q.live[0] = PB.q.live[0] + o; S.q.live[1] = PB.q.live[1] + o; S.q.live_cur = 0;
    S.q.gs.keys = PB.q.gs.keys + o; S.q.gs.keys2 = PB.q.gs.keys2 + o; S.q.gs.sorted = PB.q.gs.sorted + o;
    S.q.gs.hist = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(PB.q.gs.hist) + (h ? (((size_t)4 * (256 * 1024 + 256) + 255) & ~(size_t)255) : 0));
    S.prd = PB.prd + 5 * o;
    for (int k = 0; k < 2; k++) { S.pos[k] = PB.pos[k] + 3 * o; S.rd[k] = PB.rd[k] + 3 * o; S.n[k] = PB.n[k] + 3 * o; S.occ[k] = PB.occ[k] + o; }
    S.kd = PB.kd + 3 * o; S.rm = PB.rm + 2 * o;
    S.cb = PB.cb + 9 * (size_t)nb * o; S.maskb = PB.maskb + (size_t)nb * o;
    return S;
}
static int pt_batch_size() {   // MIRRES_PT_BATCH = samples per batch (default 64 since the end of round 6: ~164 M slots, ~110 GB of pool at 1600^2 — the device has 288 GB, and
                               // carve_batch halves the batch when it cannot have them; 32, the default of rounds 1-6: 512-spp frame -0.3 % / -1.2 %, profiles/r06_ab_pt_batch.txt;
                               // 1 = sample by sample)
    const char* e = getenv("MIRRES_PT_BATCH");   // read per frame (tests switch it)
    int k = e ? atoi(e) : 64; if (k < 1) k = 1; if (k > 64) k = 64;
    return k;
}
// The bulk stream at the LOWEST priority (MIRRES_BULK_PRIO=low): the sample-by-sample chain on the caller's stream is what bounds a small frame (a strip of a
// multi-GPU frame: profiles/r05_strip_table.txt), and its small launches get the CUs first while the batched stages fill in behind them. Default: normal priority.
static hipError_t create_side_stream(hipStream_t* out) {
    static const int prio_mode = [] { const char* e = getenv("MIRRES_BULK_PRIO"); return (e && e[0] == 'l') ? 1 : 0; }();
    if (prio_mode == 1) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) return hipStreamCreateWithPriority(out, hipStreamNonBlocking, least);
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
static int stream_count() { const char* e = getenv("MIRRES_STREAMS"); const int n = e ? atoi(e) : 2; return n < 1 ? 1 : (n > 5 ? 5 : n); }   // 1: everything on the caller's stream; 2 (default since the end of round 4: equal on the icosphere, +0.7 % on the lego-like mesh, profiles/r04_ab_gs_bits.txt): + one bulk stream; 3 (rounds 1-4): + path tracing on its own; 4: + final stages on their own; 5: + a second path-tracing stream (4, 5: measured within noise of 3)
static int carve_batch(mirres_ctx* ctx, int N, int K, int max_bounce, size_t TS, PtBatch& PB) {
    if (K < 1) K = 1;
    while ((size_t)K * (size_t)N > 0x30000000ull && K > 1) K--;   // slot indices are 32-bit
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int nb = max_bounce > 0 ? max_bounce : 1;
    auto bytes_for = [&](int k) {
        const size_t NV = (size_t)k * (size_t)N;
        return al(sizeof(Ray) * 2 * NV) + al(4 * 2 * NV) + al(sizeof(Ray) * NV) + al(sizeof(HitRec) * NV) + al(64) + 5 * al(4 * NV) + al(4 * 18 * NV)
             + al(4 * 5 * NV) + 2 * (3 * al(4 * 3 * NV) + al(4 * NV)) + al(4 * 3 * NV) + al(4 * 2 * NV) + al(4 * 9 * NV * (size_t)nb) + al(4 * NV * (size_t)nb)
             + 4 * al(4 * 8 * NV) + al(4 * 3 * (size_t)k * TS) + al(4 * (size_t)k * TS) + al(32 * (size_t)k * TS)
             + 2 * (al(sizeof(Ray) * NV) + 2 * al(4 * NV) + al(64))
             + 3 * al(4 * NV) + 2 * al(4 * (256 * 1024 + 256));      // radix sort in front of the material lookup: two key buffers + second list, two sets of digit counters (one per half)
    };
    // A request the device could not hold last time is not repeated every frame (a failing multi-GB hipMalloc per frame): the batch size that fitted is
    // remembered and later requests are clamped to it. The clamp is not for ever — a transient shortage (another tenant of the HBM) must not pin a long
    // run to small batches — but a retry is SPECULATIVE: after `ptb_retry_wait` clamped frames, and only if hipMemGetInfo says the full request fits
    // BESIDE the working pool (and under MIRRES_POOL_LIMIT_MB), the larger pool is allocated next to the working one, which is given up only once the
    // new one exists. A retry that fails anyway (fragmentation) doubles the wait (64, 128, ... 4096 frames), so a device that stays full costs one
    // failed hipMalloc per ever-longer interval, never a synchronisation, a free or a memset of the working pool.
    const char* lim_s0 = getenv("MIRRES_POOL_LIMIT_MB"); const size_t lim0 = lim_s0 ? (size_t)atoll(lim_s0) << 20 : 0;
    if (ctx->ptb_kcap > 0 && K > ctx->ptb_kcap) {
        const size_t want = bytes_for(K);
        bool grown = false;
        if (++ctx->ptb_kcap_age >= ctx->ptb_retry_wait) {
            ctx->ptb_kcap_age = 0;
            size_t fr = 0, tot = 0;
            const bool roomy = !(lim0 && want > lim0) && hipMemGetInfo(&fr, &tot) == hipSuccess && fr > want + (1ull << 30);
            char* fresh = nullptr;
            if (roomy && hipMalloc(&fresh, want) == hipSuccess) {
                if (ctx->ptb) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipFree(ctx->ptb)); }
                ctx->ptb = fresh; ctx->ptb_bytes = want;
                MR_HIP(hipMemset(ctx->ptb, 0, want));
                ctx->ptb_kcap = 0; ctx->ptb_retry_wait = 64; grown = true;
            } else {
                (void)hipGetLastError();
                ctx->ptb_retry_wait = ctx->ptb_retry_wait >= 2048 ? 4096 : 2 * ctx->ptb_retry_wait;
            }
        }
        if (!grown) K = ctx->ptb_kcap;
    }
    size_t need = bytes_for(K);
    if (ctx->ptb_bytes < need) {
        // ~670 bytes per slot: 55 GB for 32 samples of a 1600^2 frame. The larger pool is allocated BEFORE the working one is given up, so that a
        // failure leaves the context with the pool it had; only when both do not fit together is the old one freed first. When the device cannot
        // spare the request at all (other tenants of the HBM), the batch is halved until it fits.
        char* fresh = nullptr;
        // MIRRES_POOL_LIMIT_MB: refuse larger pools as if the device were full (how the tests exercise the fall-back without filling 288 GB)
        const char* lim_s = getenv("MIRRES_POOL_LIMIT_MB"); const size_t lim = lim_s ? (size_t)atoll(lim_s) << 20 : 0;
        auto pool_alloc = [&](char** p, size_t b) -> hipError_t { if (lim && b > lim) return hipErrorOutOfMemory; return hipMalloc(p, b); };
        hipError_t e = pool_alloc(&fresh, need);
        if (e != hipSuccess) {
            (void)hipGetLastError(); fresh = nullptr;
            if (e != hipErrorOutOfMemory) { set_error("mirres_render: cannot allocate the %zu-byte batch pool (%s)", need, hipGetErrorString(e)); return MIRRES_E_HIP; }
            if (ctx->ptb) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipFree(ctx->ptb)); ctx->ptb = nullptr; ctx->ptb_bytes = 0; }
            for (;;) {
                e = pool_alloc(&fresh, need);
                if (e == hipSuccess) break;
                (void)hipGetLastError(); fresh = nullptr;
                if (e != hipErrorOutOfMemory || K == 1) {   // nothing fits now: no pool, no remembered size — the next frame starts over
                    ctx->ptb_kcap = 0; set_error("mirres_render: cannot allocate the %zu-byte batch pool (%s)", need, hipGetErrorString(e)); return MIRRES_E_HIP;
                }
                K = (K + 1) / 2; need = bytes_for(K);
                ctx->ptb_kcap = K;
            }
        } else if (ctx->ptb) { MR_HIP(hipDeviceSynchronize()); MR_HIP(hipFree(ctx->ptb)); ctx->ptb = nullptr; ctx->ptb_bytes = 0; }
        ctx->ptb = fresh;
        MR_HIP(hipMemset(ctx->ptb, 0, need));   // slots that never receive a vertex are read (and ignored) by the bounce kernels
        ctx->ptb_bytes = need;
    }
    const size_t NV = (size_t)K * (size_t)N;
    char* p = ctx->ptb;
    auto take = [&](size_t b) { char* r = p; p += al(b); return r; };
    PB.K = K;
    PB.q.any_rays = (Ray*)take(sizeof(Ray) * 2 * NV); PB.q.any_hit = (int32_t*)take(4 * 2 * NV);
    PB.q.cl_rays = (Ray*)take(sizeof(Ray) * NV); PB.q.cl_hit = (HitRec*)take(sizeof(HitRec) * NV);
    PB.q.counters = (uint32_t*)take(64);
    PB.q.slot_a = (int32_t*)take(4 * NV); PB.q.mask_a = (uint32_t*)take(4 * NV); PB.q.slot_c = (int32_t*)take(4 * NV);
    PB.q.live[0] = (int32_t*)take(4 * NV); PB.q.live[1] = (int32_t*)take(4 * NV); PB.q.live_cur = 0;
    PB.q.pend = (float*)take(4 * 18 * NV);
    PB.q.N = N; PB.q.NV = (int)NV; PB.q.first_sample_is_zero = 0; PB.q.lane = 0;
    PB.prd = (float*)take(4 * 5 * NV);
    for (int k = 0; k < 2; k++) { PB.pos[k] = (float*)take(4 * 3 * NV); PB.rd[k] = (float*)take(4 * 3 * NV); PB.n[k] = (float*)take(4 * 3 * NV); PB.occ[k] = (float*)take(4 * NV); }
    PB.kd = (float*)take(4 * 3 * NV); PB.rm = (float*)take(4 * 2 * NV);
    PB.cb = (float*)take(4 * 9 * NV * (size_t)nb);
    PB.maskb = (uint32_t*)take(4 * NV * (size_t)nb);
    for (int k = 0; k < 4; k++) {
        mirres_res_t& r = (k < 2) ? PB.rinit[k] : PB.rspat[k - 2];
        r.light_data = (float*)take(4 * 8 * NV); r.light_pdf = nullptr; r.M = nullptr; r.weight = nullptr;
    }
    PB.tile_data = (float*)take(4 * 3 * (size_t)K * TS); PB.tile_pdf = (float*)take(4 * (size_t)K * TS); PB.tile_aux = (float*)take(32 * (size_t)K * TS);
    PB.q.gs.keys = (uint32_t*)take(4 * NV); PB.q.gs.keys2 = (uint32_t*)take(4 * NV); PB.q.gs.sorted = (int32_t*)take(4 * NV);
    PB.q.gs.hist = (uint32_t*)take(4 * (256 * 1024 + 256)); (void)take(4 * (256 * 1024 + 256));
    PB.qf = PB.q;
    PB.qf.any_rays = (Ray*)take(sizeof(Ray) * NV); PB.qf.any_hit = (int32_t*)take(4 * NV); PB.qf.slot_a = (int32_t*)take(4 * NV); PB.qf.counters = (uint32_t*)take(64);
    PB.qf.cl_rays = nullptr; PB.qf.cl_hit = nullptr; PB.qf.mask_a = nullptr; PB.qf.slot_c = nullptr; PB.qf.pend = nullptr; PB.qf.live[0] = PB.qf.live[1] = nullptr; PB.qf.live_cur = 0;
    PB.qv = PB.qf;
    PB.qv.any_rays = (Ray*)take(sizeof(Ray) * NV); PB.qv.any_hit = (int32_t*)take(4 * NV); PB.qv.slot_a = (int32_t*)take(4 * NV); PB.qv.counters = (uint32_t*)take(64);
    return 0;
}

// ---- band pipeline of the per-sample chain (round 6)
// The temporal -> spatial chain is the one dependent sequence of a frame: sample i + 1 needs sample i. The dependence is LOCAL, though: the spatial pass of a pixel reads
// the (temporal output) reservoirs of pixels within gather_radius = 30 px (SpatialResampling.slang:33-39), and with the temporal merge of sample i + 1 fused into the
// resolve of sample i, a band of rows of sample i + 1 needs sample i of the band itself and of its two neighbours only. The frame is therefore cut into B bands of rows;
// unit (i, j) = generate + trace + resolve of sample i on band j; unit (i + 1, j) waits for unit (i, j + 1) — which, the units of a sample running in order on one
// stream, implies (i, j) and (i, j - 1) — and the samples alternate between S chain streams with queue resources of their own (ChainSet): S samples are in flight, a band
// or two apart. A full 1600 x 1600 frame fills the device with one sample's launches (B = 1 there, nothing changes); a small frame — the 800 x 800 training frame, a
// strip — does not: its three dependent launches per sample last as long as their longest rays, and two or three of them side by side fill the idle CUs.
// Exact for any B and S: every pixel's generator, rays and merge are what the single launch computes (the order of the rays in a queue is the only thing that changes,
// and one row per band is generated and traced twice: the fused temporal merge of a band's last row may recompute the pixel below it, whose rays must be in the unit's
// own queue).
// MEASURED (profiles/r06_ab_bands.txt) AND OFF BY DEFAULT: bit-identical for every B and S tried (tests/test_gpu_fullsize.py), and slower everywhere — the 800 x 800
// training step 30.1 -> 33.4 .. 37.7 ms, an 800 x 800 frame of 128 spp 73.7 -> 83.7 .. 88.2 ms, the full frame -8 % / -5 %. Two reasons. (1) The premise is weaker
// than it looked: an 800 x 800 frame takes 576 us per sample against 422 us for a quarter of the full frame's sample — the idle share the pipeline could fill is 27 %, not
// the factor the chain's share of the summed kernel time suggested (the batched stages run beside the chain and slow it: the device is mostly busy). (2) A unit is three
// dependent launches whose latency does not shrink with the band: a traversal launch lasts as long as its longest rays (~100 us) however few it has, so B bands cost
// B tails per sample and stream where the single launch pays one; two or three streams overlap them but do not remove them. What would be needed is ONE resident
// traversal kernel that takes band queues as they become ready — a different engine. MIRRES_BANDS = n switches the pipeline on.
static int band_count(const mirres_ctx* ctx, bool allowed) {   // MIRRES_BANDS: unset / 0 / 1 = off (default), n = n bands (each at least 48 rows, boundaries at multiples of 16 rows)
    if (!allowed) return 1;
    const char* e = getenv("MIRRES_BANDS"); int b = e ? atoi(e) : 1;
    if (b <= 0) b = 1;
    const int most = ctx->fy / 48;
    if (b > most) b = most; if (b > 16) b = 16; if (b < 1) b = 1;
    return b;
}
static int chain_stream_count() { const char* e = getenv("MIRRES_CHAIN_STREAMS"); const int n = e ? atoi(e) : 2; return n < 1 ? 1 : (n > 3 ? 3 : n); }
static int ensure_chain_sets(mirres_ctx* ctx, int S) {
    ChainSet& c0 = ctx->chain_sets[0];
    c0.q = ctx->any_rays; c0.hit = ctx->any_hit; c0.counter = &ctx->counters[0]; c0.slot = ctx->slot_a; c0.mask = ctx->mask_a; c0.head_set = 0;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t N = ctx->N, pairs = ctx->any_cap / 2;
    for (int t = 1; t < S; t++) {
        if (!ctx->chain_streams[t - 1]) MR_HIP(hipStreamCreateWithFlags(&ctx->chain_streams[t - 1], hipStreamNonBlocking));
        if (!ctx->chain_mem[t - 1]) {
            const size_t bytes = al(8 * pairs) + al(4 * ctx->any_cap) + al(4 * N) + al(4 * N) + 256;      // pixel pairs, hit bits, slot, mask, counter
            char* p = nullptr; MR_HIP(hipMalloc(&p, bytes)); MR_HIP(hipMemset(p, 0, bytes));
            ctx->chain_mem[t - 1] = p;
            ChainSet& c = ctx->chain_sets[t];
            c.q = reinterpret_cast<Ray*>(p); p += al(8 * pairs); c.hit = reinterpret_cast<int32_t*>(p); p += al(4 * ctx->any_cap);
            c.slot = reinterpret_cast<int32_t*>(p); p += al(4 * N); c.mask = reinterpret_cast<uint32_t*>(p); p += al(4 * N); c.counter = reinterpret_cast<uint32_t*>(p);
            c.head_set = 16 + t;
        }
    }
    for (int t = 0; t < 3; t++) ctx->chain_sets[t].clean = false;
    return 0;
}

static int finish(mirres_ctx* ctx, const mirres_render_args_t* a, float* tot[6], FrameBufs& B, hipStream_t s) {
    const int N = (int)ctx->N; const size_t n3 = 3 * (size_t)N;
    const int spp = a->spp;
    k_average<<<grid_for(n3, MR_BLOCK), MR_BLOCK, 0, s>>>(n3, (float)spp, tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], B.comb);
    const float* srcs[5] = {tot[1], tot[2], B.comb, tot[4], tot[5]};
    if (a->gb_depth) {
        // bilateral_denoiser(_no_di) with factor 2 (renderer_restir.py:529-541): sigma = max(2 * factor, 1e-4); the packed tap records live in
        // the (now idle) packed G-buffer area of the pool (16 floats per pixel >= the 8 needed)
        const float sigma = fmaxf(2.0f * 2.0f, 0.0001f);
        float* const dsts[5] = {a->outs[1], a->outs[2], a->outs[3], a->outs[4], a->outs[5]};
        int rc = launch_bilateral5(ctx->fx, ctx->fy, sigma, srcs, a->normal, a->gb_depth, B.bil, dsts, s); if (rc) return rc;   // one pass: the five buffers share the weights
    } else {
    // EAWDenoise_use_phi(_no_di) (Denoising.py:154-251): stepWidth, then stepWidth/2, ... — the five buffers go through each iteration together
    // (k_eaw5); intermediate results alternate between a scratch set (the idle bilateral area of the pool) and the outputs so that the last
    // iteration lands in the outputs
    if (a->denoise_iter <= 0) {
        for (int k = 0; k < 5; k++) MR_HIP(hipMemcpyAsync(a->outs[k + 1], srcs[k], sizeof(float) * n3, hipMemcpyDeviceToDevice, s));
    } else {
        float* scratch[5]; float* outs5[5];
        for (int k = 0; k < 5; k++) { scratch[k] = B.bil + n3 * (size_t)k; outs5[k] = a->outs[k + 1]; }
        const float* cur[5] = {srcs[0], srcs[1], srcs[2], srcs[3], srcs[4]};
        float swf = (float)a->step_width;
        for (int it = 0; it < a->denoise_iter; it++) {
            float* const* dst = ((a->denoise_iter - 1 - it) & 1) ? scratch : outs5;
            int rc = launch_eaw5(ctx->fx, ctx->fy, (int)swf, a->c_phi, a->n_phi, a->p_phi, a->occ, cur, a->normal, a->pos, dst, s);
            if (rc) return rc;
            for (int k = 0; k < 5; k++) cur[k] = dst[k];
            swf = swf / 2;
        }
    }
    }
    k_composite<<<grid_for(N, MR_BLOCK), MR_BLOCK, 0, s>>>(N, a->occ, a->kd, a->rough_metal, a->outs[1], a->outs[2], a->outs[3], a->outs[0]);
    MR_LAUNCH_CHECK("render_finish");
    return 0;
}

extern "C" {

int mirres_ctx_reserve(mirres_ctx_t* ctx, int samples_per_batch) {
    if (!ctx || samples_per_batch < 0) { set_error("mirres_ctx_reserve: bad argument"); return MIRRES_E_ARG; }
    int K = samples_per_batch > 0 ? samples_per_batch : pt_batch_size();
    if (K > 64) K = 64;
    PtBatch PB;
    const size_t TS = (size_t)ctx->cfg.light_tile_count * ctx->cfg.light_tile_size;
    const int rc = carve_batch(ctx, ctx->fx * ctx->fy, K, ctx->cfg.max_bounce, TS, PB);
    return rc ? rc : PB.K;
}

int mirres_render(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_render_args_t* a, void* stream) {
    if (!ctx || !bvh || !a || !a->env_map || !a->occ || !a->normal || !a->depth || !a->kd || !a->rough_metal || !a->ray_dir || !a->pos || a->spp <= 0) {
        set_error("mirres_render: bad argument"); return MIRRES_E_ARG;
    }
    for (int k = 0; k < 6; k++) if (!a->outs[k]) { set_error("mirres_render: outs[%d] is null", k); return MIRRES_E_ARG; }
    if (bvh->T < 2) { set_error("mirres_render: BVH not built"); return MIRRES_E_STATE; }
    if (int e = bvh_sticky_error(bvh, "mirres_render")) return e;
    hipStream_t s = (hipStream_t)stream;
    const int N = (int)ctx->N, fx = ctx->fx; const size_t n3 = 3 * (size_t)N;
    const int Wc = a->Wc, Hc = a->Hc;
    FrameBufs B; int rc = carve(ctx, Wc, Hc, B); if (rc) return rc;
    const bool strip = a->strip_full_fy > 0;
    // the local frame may extend below the image (rows the caller padded with background so that strips of different views share a context size);
    // the own rows must lie inside it
    if (strip && !(a->own_y0 >= 0 && a->own_y0 < a->own_y1 && a->own_y1 <= ctx->fy && a->strip_y_off >= 0 && a->strip_y_off + a->own_y1 <= a->strip_full_fy)) {
        set_error("mirres_render: strip rows [%d,%d) of a %d-row local frame at global row %d of %d", a->own_y0, a->own_y1, ctx->fy, a->strip_y_off, a->strip_full_fy);
        return MIRRES_E_ARG;
    }
    if (strip) {   // the halo must cover the spatial gather radius wherever the image continues beyond the own rows
        const int r = (int)ctx->cfg.gather_radius;
        const int above = a->strip_y_off + a->own_y0, below = a->strip_full_fy - (a->strip_y_off + a->own_y1);
        if (a->own_y0 < (above < r ? above : r) || ctx->fy - a->own_y1 < (below < r ? below : r)) { set_error("mirres_render: strip halo narrower than the gather radius %d", r); return MIRRES_E_ARG; }
    }
    const bool partial = strip || !(a->spp_begin == 0 && a->spp_end == 0);
    const bool sliced = !(a->spp_begin == 0 && a->spp_end == 0);
    const int i0 = sliced ? a->spp_begin : 0, i1 = sliced ? a->spp_end : a->spp;
    const int grd = grid_for(N, MR_BLOCK);

    k_prep<<<grd, MR_BLOCK, 0, s>>>(N, a->occ, a->ray_dir, a->normal, a->depth, a->kd, a->rough_metal, B.ray_dir, B.nd, B.brdf, a->pos, reinterpret_cast<float4*>(B.grec));
    k_flip_env<<<grid_for((size_t)Wc * Hc, MR_BLOCK), MR_BLOCK, 0, s>>>(Wc, Hc, a->env_map, B.tex);
    rc = mirres_env_make_sampleable(B.tex, Wc, Hc, B.pdf, B.cdf, B.mpdf, B.mcdf, s); if (rc) return rc;
    // zero-initialised state of restir_di_with_pt (:252-302)
    MR_HIP(hipMemsetAsync(B.tot[0], 0, sizeof(float) * (size_t)(B.c1 - B.tot[0]), s));       // the six running totals (the reservoirs and the path state live in the batch pool)
    mirres_env_t E = {B.tex, Wc, Hc, B.pdf, B.cdf, B.mpdf, B.mcdf};
    // strip sharding: `occ` (halo rows zeroed) selects the pixels this rank computes; the spatial pass tests neighbours against the true G-buffer
    const float* occ = a->occ;
    struct StripGuard { mirres_ctx* c; ~StripGuard() { c->y_off = 0; c->full_fy = 0; c->occ_own = nullptr; } } strip_guard{ctx};
    if (strip) {
        k_own_occ<<<grd, MR_BLOCK, 0, s>>>(N, fx, a->own_y0, a->own_y1, a->occ, B.occ_own);
        occ = B.occ_own; ctx->y_off = a->strip_y_off; ctx->full_fy = a->strip_full_fy; ctx->occ_own = B.occ_own;
    }
    mirres_gbuf_t G = {occ, a->pos, B.nd, B.brdf, B.ray_dir};          // own-pixel stages (initial, temporal)
    mirres_gbuf_t Gt = {a->occ, a->pos, B.nd, B.brdf, B.ray_dir};      // spatial reuse: neighbours in the halo rows are real pixels
    struct GrecGuard { mirres_ctx* c; ~GrecGuard() { c->grec = nullptr; c->chain_reset = false; c->chain_clean = false; c->row_mode = 0; } } grec_guard{ctx};
    ctx->grec = B.grec;   // packed copy for the neighbour gathers of the spatial merge; cleared when this call returns (the launches captured the pointer)
    const uint32_t passes = 20;  // mTotalRISPasses (:242)
    const int max_bounce = ctx->cfg.max_bounce;

    // ---- K-sample batches. The ReSTIR stages of a sample need the previous sample's reservoirs (temporal reuse) and run one sample at a
    // time; the path-tracing stages depend only on the G-buffer and the sample's RNG stream, so the K samples of a batch go through them
    // together: K * N slots per launch. A traversal launch has a tail as long as its slowest rays (~0.1 ms) during which most CUs idle —
    // a third of a 2.3 M-ray launch, a small fraction of a K-times larger one.
    if (i1 > i0) {   // an empty slice of the sample range (spp sharding with more ranks than samples) leaves the zeroed totals
    const int Kmax = pt_batch_size();
    const size_t TS = (size_t)ctx->cfg.light_tile_count * ctx->cfg.light_tile_size;
    // K: the configured batch size (or the whole sample range when that is shorter). The path-tracing stages do not depend on the ReSTIR stages,
    // so even a single batch keeps several streams busy; larger batches mean fewer, larger launches (measured at 1600^2: 16 spp in one batch
    // of 16 instead of four of 4: 53.6 -> 50.5 ms; 128 spp in batches of 16 / 32 / 64: 375 / 370 / 367 ms)
    int Kuse = Kmax;
    { const char* e = getenv("MIRRES_MIN_BATCHES");
      const int n = i1 - i0;
      // a frame whose samples fit ONE batch runs its stages strictly one after the other (initial resampling of all samples, then the whole chain, then the final
      // stage); two batches let I(1) and F(0) run beside the chain. On a small frame (the 800 x 800 x 32 spp training frame) that is worth 1.3 % of the step
      // (profiles/r06_ab_train_batch.txt: 16 per batch 25.9 ms, 32: 26.25, 11: 27.2, 8: 28.3); on the full-size frame short batches lose (r05_ab_batch_ramp.txt)
      // A full-size frame of 33 ... 64 samples (a rank's slice of the 512-spp frame on eight GPUs) keeps the two batches of <= 32 it had while 32 was the batch size.
      const int mb = e ? (atoi(e) > 0 ? atoi(e) : 1) : ((n >= 16 && n <= Kuse && ((size_t)N <= (size_t)1024 * 1024 || n > 32)) ? 2 : 1);
      if (n < mb * Kuse) Kuse = (n + mb - 1) / mb; if (Kuse < 1) Kuse = 1; if (Kuse > Kmax) Kuse = Kmax; if (Kuse > n) Kuse = n; }
    PtBatch PB; rc = carve_batch(ctx, N, Kuse, max_bounce, TS, PB); if (rc) return rc;
    // ---- schedule. Per batch b of K samples:
    //   I(b)  initial resampling of the K samples (light tiles, candidates, shadow rays)              bulk stream, K * N slots per launch
    //   C(b)  temporal + spatial reuse, one sample after the other (needs the previous sample)        caller's stream, N pixels per launch
    //   F(b)  final visibility + evaluation + shading of the K samples -> totals 0..2                 bulk stream
    //   PT(b) new direction + max_bounce indirect vertices of the K samples -> totals 3..5            path-tracing stream
    // The branches share only read-only inputs (G-buffer, environment tables, BVH). The large launches of the bulk and path-tracing streams
    // fill the CUs the small sample-by-sample launches of the chain leave idle (a 2.3 M-ray traversal launch idles a third of the chip in its
    // tail), and the launch gaps and kernel tails of one stream are covered by the others. Reservoir sets alternate with the batch parity;
    // hand-offs are events:
    //   bulk:   wait C(b-1) | F(b-1) | I(b+1) | signal          chain:  wait signal(b-1) | C(b) | signal          path tracing: PT(0) PT(1) ...
    // PT(b) reads nothing the ReSTIR stages write (its own rays, queues and totals 3..5), so only the frame's start and end order it against
    // them. Every stream works on its own traversal head set (bvh_trace.hip) and no kernel accumulates across streams, so the frame is
    // bit-identical for any stream count and batch size (tests/test_gpu_fullsize.py). MIRRES_STREAMS=2 puts PT(b) behind I(b+1) on the bulk
    // stream; instrumented frames (counters / per-launch event timing) and MIRRES_STREAMS=1 run the same sequence on one stream.
    hipStream_t sp = s, st = s, sf = s, st2 = nullptr;   // sp: I stages, sf: F stages, st (and st2): path-tracing stages
    unsigned long long* d_sums = nullptr;
    // Whatever way this call returns, the caller's stream is ordered after every side stream that was forked from it (an error exit must not
    // leave work running on the side streams that the caller's next enqueue could race with), and the debug buffer is released.
    struct Join {
        mirres_ctx* c; hipStream_t s; hipStream_t *sp, *st, *sf, *st2; unsigned long long** sums; bool done;
        void run() {
            if (done) return; done = true;
            if (*sp != s && c->ev_join) { (void)hipEventRecord(c->ev_join, *sp); (void)hipStreamWaitEvent(s, c->ev_join, 0); }
            if (*st != *sp && c->ev_join_pt) { (void)hipEventRecord(c->ev_join_pt, *st); (void)hipStreamWaitEvent(s, c->ev_join_pt, 0); }
            if (*sf != *sp && c->ev_join_fin) { (void)hipEventRecord(c->ev_join_fin, *sf); (void)hipStreamWaitEvent(s, c->ev_join_fin, 0); }
            if (*st2 && c->ev_join_pt2) { (void)hipEventRecord(c->ev_join_pt2, *st2); (void)hipStreamWaitEvent(s, c->ev_join_pt2, 0); }
        }
        ~Join() { run(); if (*sums) { (void)hipStreamSynchronize(s); (void)hipFree(*sums); *sums = nullptr; } }
    } join{ctx, s, &sp, &st, &sf, &st2, &d_sums, false};
    const int nstreams = ctx->instrument == 0 ? stream_count() : 1;
    const bool two_streams = nstreams >= 2;
    const int nbatch = (i1 - i0 + PB.K - 1) / PB.K;
    if (two_streams) {
        if (!ctx->aux_stream) {
            MR_HIP(create_side_stream(&ctx->aux_stream));
            MR_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)); MR_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        while ((int)ctx->ev_sync.size() < 3 * (nbatch + 1)) { hipEvent_t e; MR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->ev_sync.push_back(e); }
        sp = st = sf = ctx->aux_stream;
        MR_HIP(hipEventRecord(ctx->ev_fork, s)); MR_HIP(hipStreamWaitEvent(sp, ctx->ev_fork, 0));
        PB.q.lane = PB.qf.lane = PB.qv.lane = 1;
        if (nstreams >= 3) {   // the path-tracing stages read only the G-buffer: nothing orders them against the ReSTIR stages but the frame's start and end
            if (!ctx->pt_stream) { MR_HIP(hipStreamCreateWithFlags(&ctx->pt_stream, hipStreamNonBlocking)); MR_HIP(hipEventCreateWithFlags(&ctx->ev_join_pt, hipEventDisableTiming)); }
            st = ctx->pt_stream;
            MR_HIP(hipStreamWaitEvent(st, ctx->ev_fork, 0));
            PB.q.lane = 2;
        }
        if (nstreams >= 5 && PB.K >= 2) {   // two path-tracing streams, each advancing half-batches (+1 % at 128 spp, -4 % at 16 spp: not the default)
            if (!ctx->pt_stream2) { MR_HIP(hipStreamCreateWithFlags(&ctx->pt_stream2, hipStreamNonBlocking)); MR_HIP(hipEventCreateWithFlags(&ctx->ev_join_pt2, hipEventDisableTiming)); }
            st2 = ctx->pt_stream2;
            MR_HIP(hipStreamWaitEvent(st2, ctx->ev_fork, 0));
        }
        if (nstreams >= 4) {
            if (!ctx->fin_stream) { MR_HIP(hipStreamCreateWithFlags(&ctx->fin_stream, hipStreamNonBlocking)); MR_HIP(hipEventCreateWithFlags(&ctx->ev_join_fin, hipEventDisableTiming)); }
            sf = ctx->fin_stream;
            MR_HIP(hipStreamWaitEvent(sf, ctx->ev_fork, 0));
            PB.qv.lane = 3;
        }
    }
    ctx->chain_reset = two_streams; ctx->chain_clean = false;   // with the other stages on their own streams and work heads, the chain's spatial passes clean up after themselves
    auto ev_bulk = [&](int b) { return ctx->ev_sync[3 * (b + 1)]; };       // bulk stream reached "I(b+1) done" in iteration b (b = -1: I(0))
    auto ev_chain = [&](int b) { return ctx->ev_sync[3 * (b + 1) + 1]; };  // chain C(b) done
    auto ev_fin = [&](int b) { return ctx->ev_sync[3 * (b + 1) + 2]; };    // F(b) done (only when the final stages have their own stream)
    auto batch_k = [&](int b) { const int ib = i0 + b * PB.K; return (i1 - ib < PB.K) ? (i1 - ib) : PB.K; };
    auto initial = [&](int b) -> int {
        const int ib = i0 + b * PB.K;
        PtQueues Q = PB.qf; Q.NV = batch_k(b) * N;
        return launch_initial_batch(ctx, bvh, &E, &G, &PB.rinit[b & 1], PB.tile_data, PB.tile_pdf, PB.tile_aux, a->random_offset + passes * (uint32_t)ib, batch_k(b), &Q, sp);
    };
    const bool dbg_sum = getenv("MIRRES_DBG_SUM") != nullptr;
    int n_sums = 0;
    if (dbg_sum) { MR_HIP(hipMalloc(&d_sums, 8 * 4096)); MR_HIP(hipMemsetAsync(d_sums, 0, 8 * 4096, s)); }
    auto csum = [&](const void* p, size_t words) { if (dbg_sum && n_sums < 4096) k_checksum<<<1024, MR_BLOCK, 0, s>>>((const uint32_t*)p, words, d_sums + n_sums++); };
    int pt_seq = 0;   // running index of the path-tracing sub-batches
    // band pipeline of the chain: only inside the streamed schedule (own work heads, packed reservoirs), not for strips (their per-sample halo exchange is a barrier over
    // the whole local frame), not for the checksum / counting / event-timing modes
    // the per-sample halo exchange of a strip: the library's own RCCL send / receive group (halo_comm) or the caller's callback
    const bool has_halo = a->halo || a->halo_comm;
    ctx->ev_halo_t_used = 0;
    auto exchange = [&](float* records, int sample, hipStream_t hs) -> int {
        if (a->halo_comm) {
            if (a->halo_n < 0 || a->halo_n > 2) { set_error("mirres_render: halo_n = %d", a->halo_n); return MIRRES_E_ARG; }
            hipEvent_t *e0 = nullptr, *e1 = nullptr;
            if (a->halo_time_stride > 0 && sample % a->halo_time_stride == 0) {
                if (ctx->ev_halo_t_used + 2 > ctx->ev_halo_t.size()) { const size_t old_n = ctx->ev_halo_t.size(); ctx->ev_halo_t.resize(old_n + 64); for (size_t q = old_n; q < ctx->ev_halo_t.size(); q++) MR_HIP(hipEventCreate(&ctx->ev_halo_t[q])); }
                e0 = &ctx->ev_halo_t[ctx->ev_halo_t_used]; e1 = &ctx->ev_halo_t[ctx->ev_halo_t_used + 1]; ctx->ev_halo_t_used += 2;
                MR_HIP(hipEventRecord(*e0, hs));
            }
            const int rce = comm_exchange_halos(a->halo_comm, records, fx, a->halo_n, a->halo_peer, a->halo_send0, a->halo_send1, a->halo_recv0, a->halo_recv1, hs);
            if (e1) MR_HIP(hipEventRecord(*e1, hs));
            return rce;
        }
        if (a->halo(a->halo_user, records, sample, (void*)hs)) { set_error("mirres_render: halo exchange callback failed at sample %d", sample); return MIRRES_E_STATE; }
        return 0;
    };
    const int nbands = band_count(ctx, two_streams && !has_halo && !dbg_sum && ctx->instrument == 0 && !(getenv("MIRRES_FUSE_TEMPORAL") && getenv("MIRRES_FUSE_TEMPORAL")[0] == '0') &&
                                       !(getenv("MIRRES_SPATIAL_RAYS") && getenv("MIRRES_SPATIAL_RAYS")[0] == '1'));
    const int nchain = nbands > 1 ? chain_stream_count() : 1;
    hipStream_t cstream[3] = {s, nullptr, nullptr};
    int band_seq = 0; bool have_last_unit = false; hipEvent_t last_unit_ev = nullptr; hipStream_t last_unit_stream = s;
    if (nbands > 1) {
        rc = ensure_chain_sets(ctx, nchain); if (rc) return rc;
        for (int t = 1; t < nchain; t++) { cstream[t] = ctx->chain_streams[t - 1]; MR_HIP(hipStreamWaitEvent(cstream[t], ctx->ev_fork, 0)); }
    }
    if (st2) while (ctx->ev_pt.size() < 2) { hipEvent_t e; MR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->ev_pt.push_back(e); }
    rc = initial(0); if (rc) return rc;
    if (two_streams) MR_HIP(hipEventRecord(ev_bulk(-1), sp));
    for (int b = 0; b < nbatch; b++) {
        const int ib = i0 + b * PB.K, kk = batch_k(b);
        // ---- bulk stream(s): F(b-1), I(b+1)
        if (b > 0) {
            if (two_streams) { MR_HIP(hipStreamWaitEvent(sp, ev_chain(b - 1), 0)); if (sf != sp) MR_HIP(hipStreamWaitEvent(sf, ev_chain(b - 1), 0)); }
            PtQueues Q = PB.qv; Q.NV = batch_k(b - 1) * N;
            rc = launch_final_batch(ctx, bvh, &E, occ, a->pos, a->normal, B.ray_dir, a->kd, a->rough_metal, &PB.rspat[(b - 1) & 1], batch_k(b - 1), &Q, B.tot[0], B.tot[1], B.tot[2],
                                    a->tape ? a->tape + 8 * (size_t)N * (size_t)((b - 1) * PB.K) : nullptr, sf);
            if (rc) return rc;
            if (sf != sp) MR_HIP(hipEventRecord(ev_fin(b - 1), sf));
        }
        if (b + 1 < nbatch) { rc = initial(b + 1); if (rc) return rc; }
        if (two_streams) MR_HIP(hipEventRecord(ev_bulk(b), sp));
        // ---- chain: temporal + spatial reuse of samples ib .. ib+kk-1
        if (two_streams) MR_HIP(hipStreamWaitEvent(s, ev_bulk(b - 1), 0));
        if (sf != sp && b >= 2) MR_HIP(hipStreamWaitEvent(s, ev_fin(b - 2), 0));   // C(b) overwrites the spatial reservoirs F(b-2) evaluates
        // The temporal merge of sample i + 1 is fused into the spatial resolve of sample i (k_spatial_resolve<., true>: same pixel, the spatial output still in
        // registers) whenever sample i + 1 belongs to the same batch — its initial reservoirs are then complete (I(b) precedes C(b)). The first sample of a batch
        // merges in a launch of its own (its initial reservoirs come from the bulk stream's I(b), which the previous batch's last resolve cannot wait for).
        static const bool fuse_temporal = [] { const char* e = getenv("MIRRES_FUSE_TEMPORAL"); return !(e && e[0] == '0'); }();
        // strip_overlap: interior rows = own rows at least gather_radius away from every strip edge that has a neighbouring rank behind it
        int in_a = a->own_y0, in_b = a->own_y1; bool overlap = false;
        if (strip && has_halo && a->strip_overlap) {
            const int r = (int)ctx->cfg.gather_radius;
            if (a->strip_y_off + a->own_y0 > 0) in_a += r;
            if (a->strip_y_off + a->own_y1 < a->strip_full_fy) in_b -= r;
            overlap = in_b > in_a;
            if (overlap && !ctx->halo_stream) {
                MR_HIP(hipStreamCreateWithFlags(&ctx->halo_stream, hipStreamNonBlocking));
                MR_HIP(hipEventCreateWithFlags(&ctx->ev_halo[0], hipEventDisableTiming)); MR_HIP(hipEventCreateWithFlags(&ctx->ev_halo[1], hipEventDisableTiming));
            }
        }
        bool merged_already = false;     // this sample's temporal merge ran inside the previous sample's resolve
        if (nbands > 1) {
            // ---- band pipeline (see band_count): the batch's samples alternate between the chain streams; unit (k, j) waits for unit (k - 1, j + 1)
            const int NBv = nbands;
            while ((int)ctx->ev_band.size() < PB.K * NBv) { hipEvent_t e; MR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->ev_band.push_back(e); }
            auto ev_unit = [&](int k, int j) { return ctx->ev_band[(size_t)k * NBv + j]; };
            auto band_row = [&](int j) { return j >= NBv ? ctx->fy : (int)(((long long)ctx->fy * j / NBv + 8) / 16 * 16); };
            for (int t = 1; t < nchain; t++) {      // the other chain streams enter the batch where the caller's stream does (I(b) done; F(b - 2) done with the reservoirs C(b) overwrites)
                if (two_streams) MR_HIP(hipStreamWaitEvent(cstream[t], ev_bulk(b - 1), 0));
                if (sf != sp && b >= 2) MR_HIP(hipStreamWaitEvent(cstream[t], ev_fin(b - 2), 0));
            }
            for (int k = 0; k < kk; k++) {
                const int i = ib + k, t = band_seq % nchain; band_seq++;
                hipStream_t cs = cstream[t];
                const uint32_t base = a->random_offset + passes * (uint32_t)i;
                uint32_t pass = 3;
                mirres_res_t rt = res_slot(PB.rinit[b & 1], k, (size_t)N), rs = res_slot(PB.rspat[b & 1], k, (size_t)N);
                if (i > 0) {
                    if (k == 0 && i > i0) {
                        // the batch's first sample merges in a launch of its own over the whole frame (its initial reservoirs come from the bulk stream): after EVERY unit of
                        // the previous sample, i.e. after its last one (the units of a sample run in order)
                        if (have_last_unit && last_unit_stream != cs) MR_HIP(hipStreamWaitEvent(cs, last_unit_ev, 0));
                        mirres_res_t rp = res_slot(PB.rspat[(b - 1) & 1], PB.K - 1, (size_t)N);
                        rc = mirres_restir_temporal(ctx, &E, &G, &G, &rt, &rp, nullptr, base + pass, cs); if (rc) return rc;
                    }
                    pass += 1;
                }
                const bool fuse_next = k + 1 < kk;
                mirres_res_t rn = res_slot(PB.rinit[b & 1], fuse_next ? k + 1 : k, (size_t)N);
                for (int j = 0; j < NBv; j++) {
                    if (k > 0 && nchain > 1) MR_HIP(hipStreamWaitEvent(cs, ev_unit(k - 1, j + 1 < NBv ? j + 1 : NBv - 1), 0));
                    const int y0 = band_row(j), y1 = band_row(j + 1);
                    SpatialBand band = {y0, y1, (j + 1 < NBv) ? y1 + 1 : y1, &ctx->chain_sets[t]};
                    rc = launch_spatial(ctx, bvh, &E, &Gt, &rs, &rt, nullptr, base + pass, cs, fuse_next ? &rn : nullptr, a->random_offset + passes * (uint32_t)(i + 1) + 3u, &band); if (rc) return rc;
                    MR_HIP(hipEventRecord(ev_unit(k, j), cs));
                }
                have_last_unit = true; last_unit_ev = ev_unit(k, NBv - 1); last_unit_stream = cs;
            }
            // the batch's chain is done when its last unit is (every earlier unit is among that unit's predecessors): the caller's stream carries the hand-off event
            if (last_unit_stream != s) MR_HIP(hipStreamWaitEvent(s, last_unit_ev, 0));
        } else
        for (int k = 0; k < kk; k++) {
            const int i = ib + k;
            const uint32_t base = a->random_offset + passes * (uint32_t)i;
            uint32_t pass = 3;
            mirres_res_t rt = res_slot(PB.rinit[b & 1], k, (size_t)N), rs = res_slot(PB.rspat[b & 1], k, (size_t)N);
            if (!merged_already) csum(rt.light_data, 8 * (size_t)N);
            if (i > 0) {
                // prev_* G-buffers alias the current ones from the second sample on (:462-465); a rank that starts in the middle of the sample
                // range (spp sharding) has no history yet and skips the merge but keeps the pass numbering
                if (i > i0 && !merged_already) {
                    mirres_res_t rp = (k > 0) ? res_slot(PB.rspat[b & 1], k - 1, (size_t)N) : res_slot(PB.rspat[(b - 1) & 1], PB.K - 1, (size_t)N);
                    rc = mirres_restir_temporal(ctx, &E, &G, &G, &rt, &rp, nullptr, base + pass, s); if (rc) return rc;
                }
                pass += 1;
            }
            if (has_halo && overlap) {
                // strip sharding with the exchange off the chain: the callback enqueues it on the side stream (after this sample's temporal output), the chain does the
                // interior rows meanwhile, then waits and does the border rows
                MR_HIP(hipEventRecord(ctx->ev_halo[0], s)); MR_HIP(hipStreamWaitEvent(ctx->halo_stream, ctx->ev_halo[0], 0));
                rc = exchange(rt.light_data, i, ctx->halo_stream); if (rc) return rc;
                MR_HIP(hipEventRecord(ctx->ev_halo[1], ctx->halo_stream));
                ctx->row_a = in_a; ctx->row_b = in_b; ctx->row_mode = 1;
                rc = launch_spatial(ctx, bvh, &E, &Gt, &rs, &rt, nullptr, base + pass, s, nullptr, 0u);
                if (!rc) { MR_HIP(hipStreamWaitEvent(s, ctx->ev_halo[1], 0)); ctx->row_mode = 2; rc = launch_spatial(ctx, bvh, &E, &Gt, &rs, &rt, nullptr, base + pass, s, nullptr, 0u); }
                ctx->row_mode = 0;
                if (rc) return rc;
                merged_already = false;
                csum(rs.light_data, 8 * (size_t)N);
                continue;
            }
            if (has_halo) {   // strip sharding: the neighbouring ranks' border rows of the temporal output -> this rank's halo rows (and vice versa)
                rc = exchange(rt.light_data, i, s); if (rc) return rc;
            }
            const bool fuse_next = fuse_temporal && !dbg_sum && k + 1 < kk;      // (sample i + 1 > i0 >= 0: it always has a temporal pass)
            mirres_res_t rn = res_slot(PB.rinit[b & 1], fuse_next ? k + 1 : k, (size_t)N);
            rc = launch_spatial(ctx, bvh, &E, &Gt, &rs, &rt, nullptr, base + pass, s, fuse_next ? &rn : nullptr, a->random_offset + passes * (uint32_t)(i + 1) + 3u); if (rc) return rc;
            merged_already = fuse_next;
            csum(rs.light_data, 8 * (size_t)N);
        }
        if (two_streams) MR_HIP(hipEventRecord(ev_chain(b), s));
        // ---- path-tracing stages of samples ib .. ib+kk-1 (new direction, then max_bounce indirect vertices), in sub-batches of Kp samples that
        // alternate between the two halves of the per-slot state (and the two path-tracing streams, when there are two)
        const int nbq = max_bounce > 0 ? max_bounce : 1;
        const int Kp = st2 ? PB.K / 2 : PB.K;
        for (int k0 = 0; k0 < kk; k0 += Kp, pt_seq++) {
            const int ks = (kk - k0 < Kp) ? (kk - k0) : Kp, is = ib + k0, h = st2 ? (pt_seq & 1) : 0;
            hipStream_t sq = h ? st2 : st;
            PtSet T = pt_set(PB, h, (size_t)Kp * (size_t)N, nbq);
            PtQueues Q = T.q; Q.NV = ks * N; Q.first_sample_is_zero = (is == 0); if (h) Q.lane = 4;
            uint32_t fi = a->random_offset + passes * (uint32_t)is + 5;   // pass number of new_dir for a sample with a temporal pass before it
            mirres_path_t P0 = {occ, a->pos, a->normal, B.ray_dir, a->kd, a->rough_metal, T.prd, T.pos[0], T.rd[0], T.occ[0], T.n[0]};
            // the bounce kernels run over live-slot lists and leave the other slots alone: the per-bounce masks k_pt_reduce reads must say "nothing" there
            if (max_bounce > 0) MR_HIP(hipMemsetAsync(T.maskb, 0, sizeof(uint32_t) * (size_t)nbq * (size_t)Q.NV, sq));
            Q.live_cur = 0;
            rc = launch_new_dir(ctx, bvh, &P0, fi, 0, sq, &Q); if (rc) return rc;
            fi += 5;
            int src = 0;
            for (int bo = 1; bo <= max_bounce; bo++) {
                // material lookup at the new vertices: compacted slot list -> hash-grid gather -> MFMA MLP -> scatter (slot_c is free between passes)
                if (a->mat && !(getenv("MIRRES_MATNET") && getenv("MIRRES_MATNET")[0] == 'v')) rc = launch_matnet_scatter_mfma(a->mat, T.occ[src], T.pos[src], Q.NV, T.kd, T.rm, a->use_scale, a->scale, Q.slot_c, &Q.counters[2], sq, Q.live[Q.live_cur], &Q.counters[3 + Q.live_cur], &Q.gs);
                else rc = launch_matnet_scatter(a->mat, T.occ[src], T.pos[src], Q.NV, T.kd, T.rm, a->use_scale, a->scale, a->const_kd, a->const_rm, sq);
                if (rc) return rc;
                mirres_path_t Pb = {T.occ[src], T.pos[src], T.n[src], T.rd[src], T.kd, T.rm, T.prd, T.pos[src ^ 1], T.rd[src ^ 1], T.occ[src ^ 1], T.n[src ^ 1]};
                float* cb = T.cb + (size_t)(bo - 1) * 9 * (size_t)Q.NV;
                Q.mask_a = T.maskb + (size_t)(bo - 1) * (size_t)Q.NV;      // kept per bounce for k_pt_reduce
                rc = launch_bounce(ctx, bvh, &E, &Pb, fi, (uint32_t)bo, cb, cb + 3 * (size_t)Q.NV, cb + 6 * (size_t)Q.NV, nullptr, nullptr, nullptr, sq, &Q); if (rc) return rc;
                fi += 5;
                src ^= 1; Q.live_cur ^= 1;
            }
            if (max_bounce > 0) {
                // totals 3..5 take the sub-batches in sample order whatever stream they ran on (fp32 sums: the order is part of the result)
                if (st2 && pt_seq > 0) MR_HIP(hipStreamWaitEvent(sq, ctx->ev_pt[(pt_seq - 1) & 1], 0));
                k_pt_reduce<<<grid_for(n3, MR_BLOCK), MR_BLOCK, 0, sq>>>(N, ks, max_bounce, T.cb, T.maskb, B.tot[3], B.tot[4], B.tot[5]);
                if (st2) MR_HIP(hipEventRecord(ctx->ev_pt[pt_seq & 1], sq));
            }
        }
    }
    {   // F(last)
        if (two_streams) MR_HIP(hipStreamWaitEvent(sf, ev_chain(nbatch - 1), 0));
        PtQueues Q = PB.qv; Q.NV = batch_k(nbatch - 1) * N;
        rc = launch_final_batch(ctx, bvh, &E, occ, a->pos, a->normal, B.ray_dir, a->kd, a->rough_metal, &PB.rspat[(nbatch - 1) & 1], batch_k(nbatch - 1), &Q, B.tot[0], B.tot[1], B.tot[2],
                                a->tape ? a->tape + 8 * (size_t)N * (size_t)((nbatch - 1) * PB.K) : nullptr, sf);
        if (rc) return rc;
    }
    join.run();
    if (dbg_sum) {
        std::vector<unsigned long long> h(n_sums);
        MR_HIP(hipStreamSynchronize(s)); MR_HIP(hipMemcpy(h.data(), d_sums, 8 * (size_t)n_sums, hipMemcpyDeviceToHost)); (void)hipFree(d_sums); d_sums = nullptr;
        for (int k = 0; k < n_sums; k++) fprintf(stderr, "[sum %d] %016llx\n", k, h[k]);
    }
    }
    if (partial) {
        for (int k = 0; k < 6; k++) MR_HIP(hipMemcpyAsync(a->outs[k], B.tot[k], sizeof(float) * n3, hipMemcpyDeviceToDevice, s));
        return MIRRES_OK;
    }
    return finish(ctx, a, B.tot, B, s);
}

int mirres_ctx_halo_time(mirres_ctx_t* ctx, double* ms, int* exchanges) {
    if (!ctx || !ms || !exchanges) { set_error("mirres_ctx_halo_time: null"); return MIRRES_E_ARG; }
    double tot = 0.0; int n = 0;
    for (size_t q = 0; q + 1 < ctx->ev_halo_t_used; q += 2) {
        MR_HIP(hipEventSynchronize(ctx->ev_halo_t[q + 1]));
        float t = 0.f; MR_HIP(hipEventElapsedTime(&t, ctx->ev_halo_t[q], ctx->ev_halo_t[q + 1]));
        tot += t; n++;
    }
    ctx->ev_halo_t_used = 0;
    *ms = tot; *exchanges = n;
    return MIRRES_OK;
}

int mirres_render_bwd(mirres_ctx_t* ctx, const mirres_render_args_t* a, int samples, const float* g_color, const float* g_diff, const float* g_spec,
                      float* g_normal, float* g_kd, float* g_rough_metal, float* g_env, void* stream) {
    if (!ctx || !a || !a->tape || samples <= 0 || !g_color || !g_diff || !g_spec || !a->env_map || !a->occ || !a->normal || !a->kd || !a->rough_metal || !a->ray_dir) {
        set_error("mirres_render_bwd: bad argument (the forward call must have recorded a tape)"); return MIRRES_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    FrameBufs B; int rc = carve(ctx, a->Wc, a->Hc, B); if (rc) return rc;
    k_flip_env<<<grid_for((size_t)a->Wc * a->Hc, MR_BLOCK), MR_BLOCK, 0, s>>>(a->Wc, a->Hc, a->env_map, B.tex);   // the map the forward sampled
    return launch_direct_bwd(B.tex, a->Wc, a->Hc, (int)ctx->N, samples, a->occ, a->normal, a->ray_dir, a->kd, a->rough_metal, a->tape, g_color, g_diff, g_spec,
                             g_normal, g_kd, g_rough_metal, g_env, s);
}

int mirres_render_finish(mirres_ctx_t* ctx, const mirres_render_args_t* a, float* sums[6], void* stream) {
    if (!ctx || !a || !sums) { set_error("mirres_render_finish: null"); return MIRRES_E_ARG; }
    for (int k = 0; k < 6; k++) if (!a->outs[k] || !sums[k]) { set_error("mirres_render_finish: null buffer %d", k); return MIRRES_E_ARG; }
    FrameBufs B; int rc = carve(ctx, a->Wc, a->Hc, B); if (rc) return rc;
    return finish(ctx, a, sums, B, (hipStream_t)stream);
}

}  // extern "C"
