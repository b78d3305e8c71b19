// passes.hip — context, environment tables, light tiles and the ReSTIR reservoir passes as wavefront kernels.
//
// Every pass that the reference runs as one megakernel with in-line traversal (InitialResampling_, SpatialResampling_,
// EvaluateFinalSamples_get_vis) is split into   generate (per pixel, emits COMPACTED shadow rays with one atomic per
// wave) -> trace (bvh_trace.hip, any-hit queue) -> resolve (per pixel, consumes hit bits).  The rays a pixel emits
// depend only on pass inputs, never on another ray's result, so the split is exact (DESIGN.md §Wavefront split).
#ifndef MR_LEAN_FP
#define MR_LEAN_FP 1      // device_math.hpp: short division / square-root sequences, bit-identical to the compiler's for the renderer's operand range
#endif
#include <algorithm>
#include "engine.hpp"
#include "device_math.hpp"
#include "device_light.hpp"
#include "device_brdf.hpp"

namespace mr {

int trace_any_queue_counted(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit,
                            unsigned long long* stats, hipStream_t s, int reference_order);

#define MR_BLOCK 256
// Block sizes of the ray-generating kernels. They are long, register-heavy (78-80 VGPRs: six waves per SIMD) per-thread programs, and a block is placed
// only when a CU has room for ALL its waves: with 1024 threads (four waves per SIMD per block) a CU held ONE block — four of the six possible waves,
// and nothing at all beside another stream's block. 512 / 256 threads fill the six: k_initial_gen 10.3 -> 7.3 ms per 32-sample batch, k_bounce_gen
// 4.9 -> 4.0 ms, frame 974 -> 1029 Msamples/s (round 2; per-kernel sizes picked on the frame: 256 / 512 / 512).
#ifndef MR_GEN_BLOCK
#define MR_GEN_BLOCK 512    // ray-generating kernels: one queue atomic per block
#endif
#ifndef MR_IGEN_BLOCK
#define MR_IGEN_BLOCK 256
#endif
#ifndef MR_SGEN_TILE
#define MR_SGEN_TILE 16
#endif
#define MR_SGEN_BLOCK (MR_SGEN_TILE * MR_SGEN_TILE)  // k_spatial_gen: one square pixel tile per block

// 2-D tile -> pixel mapping for the kernels that gather from neighbouring pixels (spatial pass: +-30 px): a square tile of threads touches a
// (T+60)^2 window instead of the (blockDim+60) x 61 strip a row-major block touches, which is what the L1/L2 hit rate of the gathers follows.
#ifndef MR_TILE_MAP
#define MR_TILE_MAP 2
#endif
#ifndef MR_CHUNK_PX
#define MR_CHUNK_PX 128
#endif
#ifndef MR_SRES_TILE
#define MR_SRES_TILE 8      // one wave per workgroup. 16 (rounds 2-4) was the better shape while the kernel held 140 registers without the fused temporal merge; at 159 (three waves
                            // per SIMD) the finer granularity wins: +2.2 % (icosphere) / +0.7 % (lego-like) on the frame (profiles/r04_ab_spatial_shapes.txt)
#endif
// MR_TILE_MAP 0: row-major tiles. 1: eight contiguous bands of tile rows, one per XCD. 2: 128 x 128 px chunks of tiles dealt to the XCDs in turn.
// (Workgroups go to the eight XCDs round-robin by block index and every XCD has its own L2.)
MR_DEV int tile_pixel(int fx, int fy, int tw, int N) {
    const int tiles_x = (fx + tw - 1) / tw, tiles_y = (fy + tw - 1) / tw;
    int tx, ty;
#if MR_TILE_MAP == 0
    tx = blockIdx.x % tiles_x; ty = blockIdx.x / tiles_x;
#elif MR_TILE_MAP == 1
    const int t = (int)(blockIdx.x & 7u) * ((tiles_x * tiles_y + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (t >= tiles_x * tiles_y) return N;
    tx = t % tiles_x; ty = t / tiles_x;
#else
    const int ch = MR_CHUNK_PX / tw, chunks_x = (tiles_x + ch - 1) / ch;
    const int j = (int)(blockIdx.x >> 3), chunk = (int)(blockIdx.x & 7u) + 8 * (j / (ch * ch)), w = j % (ch * ch);
    tx = (chunk % chunks_x) * ch + w % ch; ty = (chunk / chunks_x) * ch + w / ch;
    if (tx >= tiles_x || ty >= tiles_y) return N;
#endif
    const int x = tx * tw + (int)(threadIdx.x % tw), y = ty * tw + (int)(threadIdx.x / tw);
    return (x < fx && y < fy) ? y * fx + x : N;   // N = "no pixel"
}
static int tile_grid(int fx, int fy, int tw) {
    const int tiles_x = (fx + tw - 1) / tw, tiles_y = (fy + tw - 1) / tw;
#if MR_TILE_MAP == 0
    return tiles_x * tiles_y;
#elif MR_TILE_MAP == 1
    return 8 * ((tiles_x * tiles_y + 7) / 8);
#else
    const int ch = MR_CHUNK_PX / tw, chunks = ((tiles_x + ch - 1) / ch) * ((tiles_y + ch - 1) / ch);
    return 8 * ((chunks + 7) / 8) * ch * ch;
#endif
}

// G-buffer / reservoir views. The ABI layout is the reference's SoA (one array per field). mirres_render's internal buffers use packed records so
// that a neighbour gather touches one or two cache lines instead of seven:  GBufD::rec = 64 B per pixel {n.xyz depth | ray_dir.xyz occ | brdf.xyz 0 |
// pos.xyz 0};  ResD::rec = 32 B per slot {light_data.xyz lum | M(int bits) weight vcode inv_pdf} (lum in the first half: the shadow-ray kernel reads it with the direction). Same values, same arithmetic — only the addresses differ.
// lum (packed records only, round 3) = luminance of the environment radiance along the stored light sample, luminance(env_radiance(E, oct_decode(light_data.yz))):
// a pure function of the sample, evaluated once when the sample enters a reservoir (the initial pass has it anyway) and carried with it through the temporal and
// spatial merges, instead of being recomputed (acos, atan2, sin, four texel gathers) by every pass that evaluates a target function for the sample — the temporal
// merge thrice, the spatial merge six times per pixel. The target function only ever needs that luminance (res.slang:70-77), and a carried value has the bits of a
// recomputed one.  Empty reservoirs (weight 0) carry 0: every product their luminance enters is multiplied by that weight.
struct GBufD { const float *occ, *pos, *normal_depth, *brdf, *ray_dir; const float4* rec; };
struct ResD { float* light_data; float* light_pdf; int32_t* M; float* weight; float4* rec; };
struct GPix { v3 n; float depth; v3 rd; float occ; v3 brdf; };
MR_DEV GPix load_gpix(const GBufD& G, size_t i) {   // normal+depth, ray_dir, occ, brdf of one pixel
    GPix p;
    if (G.rec) {
        const float4 a = G.rec[4 * i], b = G.rec[4 * i + 1], c = G.rec[4 * i + 2];
        p.n = V3(a.x, a.y, a.z); p.depth = a.w; p.rd = V3(b.x, b.y, b.z); p.occ = b.w; p.brdf = V3(c.x, c.y, c.z);
    } else {
        p.n = V3(G.normal_depth[4 * i], G.normal_depth[4 * i + 1], G.normal_depth[4 * i + 2]); p.depth = G.normal_depth[4 * i + 3];
        p.rd = ld3(G.ray_dir, i); p.occ = G.occ[i]; p.brdf = ld3(G.brdf, i);
    }
    return p;
}
MR_DEV v3 load_gpos(const GBufD& G, size_t i) { if (G.rec) { const float4 d = G.rec[4 * i + 3]; return V3(d.x, d.y, d.z); } return ld3(G.pos, i); }
// vcode (packed records only): what is already known about the visibility of the stored light sample FROM THIS PIXEL — 0 unknown, 1 visible,
// 2 occluded. The shadow ray that the final-visibility stage would trace for a sample is bit-identical (same origin, direction, offset) to one
// an earlier stage traced for it — the initial candidate's ray, or the spatial pass's "canonical pixel towards the neighbour's light" ray —
// so that stage's answer is carried along with the sample (through the temporal merge only when history comes from the same pixel) and the
// final stage traces only what is still unknown: one shadow ray in five disappears, results unchanged by construction.
struct ResV { v3 light_data; float light_pdf; int M; float weight; int vcode; float lum; bool has_lum; };
struct Ris { v3 light_data; float inv_pdf, weightSum, M, weight, canonicalWeight; int vcode; float lum; };

MR_DEV Ris empty_ris() { Ris s; s.light_data = V3(0.f); s.inv_pdf = 0.f; s.weightSum = 0.f; s.M = 0.f; s.weight = 0.f; s.canonicalWeight = 0.f; s.vcode = 0; s.lum = 0.f; return s; }
MR_DEV ResV load_res(const ResD& R, size_t i) {
    ResV r;
    if (R.rec) { const float4 a = R.rec[2 * i], b = R.rec[2 * i + 1]; r.light_data = V3(a.x, a.y, a.z); r.lum = a.w; r.M = __float_as_int(b.x); r.weight = b.y; r.vcode = __float_as_int(b.z); r.light_pdf = b.w; r.has_lum = true; }
    else { r.light_data = ld3(R.light_data, i); r.light_pdf = R.light_pdf[i]; r.M = R.M[i]; r.weight = R.weight[i]; r.vcode = 0; r.lum = 0.f; r.has_lum = false; }
    return r;
}
MR_DEV void store_res(const ResD& R, size_t i, v3 ld, float ipdf, int M, float w, int vcode = 0, float lum = 0.f) {
    if (R.rec) { float4 a, b; a.x = ld.x; a.y = ld.y; a.z = ld.z; a.w = lum; b.x = __int_as_float(M); b.y = w; b.z = __int_as_float(vcode); b.w = ipdf; R.rec[2 * i] = a; R.rec[2 * i + 1] = b; }
    else { st3(R.light_data, i, ld); R.light_pdf[i] = ipdf; R.M[i] = M; R.weight[i] = w; }
}
MR_DEV v3 res_light(const ResD& R, size_t i) { if (R.rec) { const float4 a = R.rec[2 * i]; return V3(a.x, a.y, a.z); } return ld3(R.light_data, i); }
MR_DEV int res_M(const ResD& R, size_t i) { return R.rec ? __float_as_int(R.rec[2 * i + 1].x) : R.M[i]; }
MR_DEV float res_weight(const ResD& R, size_t i) { return R.rec ? R.rec[2 * i + 1].y : R.weight[i]; }
MR_DEV int res_vcode(const ResD& R, size_t i) { return R.rec ? __float_as_int(R.rec[2 * i + 1].z) : 0; }
MR_DEV float res_lum(const ResD& R, size_t i) { return R.rec ? R.rec[2 * i].w : 1.f; }   // carried luminance of the stored sample (packed records; SoA: unknown, never 0)
MR_DEV void store_zero(const ResD& R, size_t i) { store_res(R, i, V3(0.f), 0.f, 0, 0.f); }
MR_DEV void store_ris(const ResD& R, size_t i, const Ris& s) {
    if (isinf(s.weight) || isnan(s.weight)) { store_zero(R, i); return; }
    store_res(R, i, s.light_data, s.inv_pdf, (int)s.M, s.weight, s.vcode, s.lum);
}
// luminance of the radiance along a loaded reservoir's sample: the carried value (packed records) or the evaluation itself (the reference's SoA layout)
MR_DEV float sample_lum(const EnvD& E, const ResV& r, v3 dir) { return r.has_lum ? r.lum : luminance(env_radiance(E, dir)); }
MR_DEV float target_lum(const rtarget::Ctx& c, float lum, v3 L) { return fmaxf(0.f, lum * rtarget::eval_brdf(c, L)); }   // rtarget::target with the luminance given
MR_DEV void put_ray(Ray* q, uint32_t slot, v3 pos, v3 dir, float vis_near, float t_max = 1e7f) {
    v3 o = pos + vis_near * dir;  // origin offset along the RAY direction (VIS_near, e.g. InitialResampling.slang:264-265)
    float4 a, b;
    a.x = o.x; a.y = o.y; a.z = o.z; a.w = 0.f;
    b.x = dir.x; b.y = dir.y; b.z = dir.z; b.w = t_max;
    reinterpret_cast<float4*>(q + slot)[0] = a; reinterpret_cast<float4*>(q + slot)[1] = b;
}
MR_DEV EnvD envd(const mirres_env_t& e) { EnvD E; E.tex = e.tex; E.W = e.Wc; E.H = e.Hc; E.pdf = e.pdf; E.cdf = e.cdf; E.mpdf = e.mpdf; E.mcdf = e.mcdf; return E; }

// ---------------------------------------------------------------- environment tables
// make_sampleable (make_sampleable.slang:34-60): weight = lum(env_le(ngp_dir(dir(h,w)))) * sin(theta)
__global__ void __launch_bounds__(MR_BLOCK) k_env_weight(const float* __restrict__ tex, int W, int H, float* __restrict__ pdf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const float PI = 3.141592653589793f;
    int h = i / W, w = i % W;
    float v = (h + .5f) / H;
    float sin_theta = mrf_sin(PI * v);
    float ux = (w + .5f) / W;
    float theta = v * PI, phi = ux * 2 * PI;
    float cos_theta, cos_phi, sin_theta_dir, sin_phi;
    mrf_sincos(theta, &sin_theta_dir, &cos_theta); mrf_sincos(phi, &sin_phi, &cos_phi);
    v3 raw = V3(sin_theta_dir * cos_phi, cos_theta, sin_theta_dir * sin_phi);
    float wv = luminance(env_le(ngp_dir(raw), tex, W, H));
    pdf[i] = wv * sin_theta;
}
// Row scans + Distribution2D (GenerateLightTiles.py:10-24, make_sampleable.slang:62-86). Sequential fp32 sums per row: a fixed summation order keeps the table
// reproducible run to run (and equal to the checker's). Round 6: the sums are the same, in the same order — but one thread per row walking global memory made the
// kernel a chain of ~3000 dependent memory operations per thread on four waves (353 us per frame for a 256 x 512 map, on the caller's stream before anything else
// can start: 1.3 % of a training step). Now one workgroup per row: the row moves through LDS in chunks (coalesced loads / stores by 64 lanes), lane 0 forms the
// running sum from LDS, every lane normalises.
#define MR_ENV_CHUNK 1024
__global__ void __launch_bounds__(64) k_env_rows(int W, int H, float* __restrict__ pdf, float* __restrict__ cdf, float* __restrict__ mpdf) {
    __shared__ float buf[MR_ENV_CHUNK];
    __shared__ float s_acc;
    const int t = threadIdx.x, y = blockIdx.x;
    if (y >= H) return;
    float* const prow = pdf + (size_t)y * W; float* const crow = cdf + (size_t)y * (W + 1);
    float acc = 0.f;                                   // lane 0's: the row's running sum
    for (int w0 = 0; w0 < W; w0 += MR_ENV_CHUNK) {
        const int n = W - w0 < MR_ENV_CHUNK ? W - w0 : MR_ENV_CHUNK;
        for (int i = t; i < n; i += 64) buf[i] = prow[w0 + i];
        __syncthreads();
        if (t == 0) for (int c = 0; c < n; c++) { acc += buf[c]; buf[c] = acc; }
        __syncthreads();
        for (int i = t; i < n; i += 64) crow[w0 + i + 1] = buf[i];
        __syncthreads();
    }
    if (t == 0) { crow[0] = 0.f; mpdf[y] = acc; s_acc = acc; }
    __syncthreads();
    const float row_weight = s_acc;
    for (int x = t; x < W; x += 64) {
        if (row_weight < 1e-4f) { prow[x] = 1.0f / W; crow[x] = x / (float)W; }
        else { prow[x] /= row_weight; crow[x] /= row_weight; }
    }
    if (t == 0) crow[W] = 1.f;
}
__global__ void __launch_bounds__(64) k_env_marginal(int H, float* __restrict__ mpdf, float* __restrict__ mcdf) {
    // one workgroup; the same sequential sum by lane 0, from LDS
    __shared__ float buf[MR_ENV_CHUNK];
    __shared__ float s_tot;
    if (blockIdx.x != 0) return;
    const int t = threadIdx.x;
    float acc = 0.f;
    for (int h0 = 0; h0 < H; h0 += MR_ENV_CHUNK) {
        const int n = H - h0 < MR_ENV_CHUNK ? H - h0 : MR_ENV_CHUNK;
        for (int i = t; i < n; i += 64) buf[i] = mpdf[h0 + i];
        __syncthreads();
        if (t == 0) for (int c = 0; c < n; c++) { acc += buf[c]; buf[c] = acc; }
        __syncthreads();
        for (int i = t; i < n; i += 64) mcdf[h0 + i + 1] = buf[i];
        __syncthreads();
    }
    if (t == 0) { mcdf[0] = 0.f; s_tot = acc; }
    __syncthreads();
    const float total = s_tot;
    for (int h = t; h < H; h += 64) mpdf[h] = mpdf[h] / total;
    for (int h = t; h <= H; h += 64) mcdf[h] = mcdf[h] / total;
    __syncthreads();
    if (t == 0) mcdf[H] = 1.f;
}

// createNeighborOffsetTexture (make_sampleable.slang:186-205) — a serial R2 walk, run once by one thread as in the reference
__global__ void k_neighbor_offsets(int count, float* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int R = 254;
    const float phi2 = 1.f / 1.3247179572447f;
    float u = 0.5f, v = 0.5f;
    for (uint32_t index = 0; index < (uint32_t)count * 2;) {
        u += phi2; v += phi2 * phi2;
        if (u >= 1.f) u -= 1.f;
        if (v >= 1.f) v -= 1.f;
        float rSq = (u - 0.5f) * (u - 0.5f) + (v - 0.5f) * (v - 0.5f);
        if (rSq > 0.25f) continue;
        out[index++] = (float)(int)((u - 0.5f) * R) / 127;
        out[index++] = (float)(int)((v - 0.5f) * R) / 127;
    }
}

// process_GenerateLightTiles (GenerateLightTiles.slang:16-62): exactly tile_count*tile_size threads (the reference
// launches 33.5 M threads of which 131 072 work).
// `total` = K * per: sample k of a K-sample batch (mirres_render) fills tiles [k * per, (k + 1) * per) with the stream of frameIndex + 20 k.
__global__ void __launch_bounds__(MR_BLOCK) k_light_tiles(EnvD E, uint32_t frameIndex, int total, int per, float* __restrict__ light_data,
                                                          int32_t* __restrict__ light_uv, float* __restrict__ light_pdf) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)total) return;
    const uint32_t ks = idx / (uint32_t)per, li = idx - ks * (uint32_t)per;
    uint32_t sg = seed_generator(li, li, frameIndex + 20u * ks + 1);  // scalar bufferIndex splat to uint2 (:34)
    float r0 = rnd(sg), r1 = rnd(sg);
    v3 ld = V3(0.f); int ux = 0, uy = 0; float ip = 0.f;
    v3 dir; float pdf; v2 luv;
    if (sample_li(E, r0, r1, dir, pdf, luv)) {
        v2 o = oct_encode(dir);
        ld = V3(1.f, o.x, o.y);
        uv2xy(luv, E.W, E.H, ux, uy);
        ip = pdf;
    }
    st3(light_data, idx, ld);
    if (light_uv) { light_uv[2 * idx] = ux; light_uv[2 * idx + 1] = uy; }
    light_pdf[idx] = ip;
}

// Per tile sample, the direction and the luminance of its radiance are functions of the sample alone (get_light_info, lightDi.slang:285-298):
// evaluated once per sample here (131 072 evaluations) instead of once per pixel x candidate in the resampling loop (82 M at 1600^2) — same
// pure functions of the same inputs, so the values are bit-identical to in-loop evaluation.
// Round 3: the record is 32 bytes — {direction xyz, luminance | pdf, light_data xyz} — everything the candidate loop of k_initial_gen reads about a tile sample,
// in one sector: two 16-byte gathers per candidate instead of five gathers from three arrays (the loop's 32 x 5 scattered loads per pixel were what it waited for).
// COMPACT (mirres_render's batches, where the tiles come from k_light_tiles above; end of round 4): 16 bytes per sample — {oct code y z, luminance, pdf}. The candidate loop
// is bound by these gathers (33 per pixel and sample out of a 4 MB table), not by arithmetic, so it re-derives what the other half of the record held: the direction
// = oct_decode of the code (the very expression evaluated here) and light_data.x, which k_light_tiles sets to 1 exactly when sample_li succeeded, i.e. when the
// stored pdf != 0 (device_light.hpp sample_li: `return !(pdf == 0)`; a failed sample is stored as all zeros). Arrays handed in through the stepwise ABI
// (mirres_restir_initial) need not obey that relation: they keep the 32-byte record.
template <bool COMPACT>
__global__ void __launch_bounds__(MR_BLOCK) k_tile_aux(EnvD E, int total, const float* __restrict__ tile_data, const float* __restrict__ tile_pdf, float4* __restrict__ aux) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    v3 ld = ld3(tile_data, idx);
    v3 ldir = oct_decode(V2(ld.y, ld.z));
    const float lum = luminance(env_radiance(E, ldir));
    if (COMPACT) { float4 c; c.x = ld.y; c.y = ld.z; c.z = lum; c.w = tile_pdf[idx]; aux[idx] = c; return; }
    float4 a; a.x = ldir.x; a.y = ldir.y; a.z = ldir.z; a.w = lum;
    float4 b; b.x = tile_pdf[idx]; b.y = ld.x; b.z = ld.y; b.w = ld.z;
    aux[2 * (size_t)idx] = a; aux[2 * (size_t)idx + 1] = b;
}

// ---------------------------------------------------------------- initial resampling (InitialResampling.slang:151-295)
template <bool COMPACT>
__global__ void __launch_bounds__(MR_IGEN_BLOCK) k_initial_gen(mirres_config_t C, EnvD E, GBufD G, ResD R, const float* __restrict__ tile_data,
                                                          const float* __restrict__ tile_pdf, const float4* __restrict__ tile_aux, uint32_t frameIndex0, int fx, int N,
                                                          int NV, int TS, int y_off, Ray* __restrict__ q, uint32_t* __restrict__ q_count, int32_t* __restrict__ slot_out) {
    // slot sv = k * N + pixel: sample k of a K-sample batch (NV = K * N; K = 1 for the stepwise ABI) — reservoir at sv, G-buffer at the pixel,
    // light tiles of sample k at k * TS, RNG stream frameIndex0 + 20 k
    const int sv = blockIdx.x * blockDim.x + threadIdx.x;
    bool want = false; v3 rpos = V3(0.f), rdir = V3(0.f);
    if (sv < NV) {
        const int ks = sv / N, pi = sv - ks * N;
        const uint32_t frameIndex = frameIndex0 + 20u * (uint32_t)ks;
        tile_aux += (COMPACT ? 1 : 2) * (size_t)ks * TS;
        const GPix gp = load_gpix(G, pi);
        if (gp.occ < 0.1f) store_zero(R, sv);
        else {
            const uint32_t x = (uint32_t)(pi % fx), y = (uint32_t)(pi / fx + y_off);   // global pixel coordinates seed the streams (strip sharding: y_off)
            uint32_t tileSg = seed_generator(x / C.screen_tile_size, y / C.screen_tile_size, frameIndex);
            uint32_t tileIndex = min((uint32_t)(rnd(tileSg) * C.light_tile_count), (uint32_t)C.light_tile_count - 1);
            uint32_t tileOffset = tileIndex * C.light_tile_size;
            uint32_t sg = seed_generator(x, y, frameIndex);
            uint32_t stride = (C.light_tile_size + C.initial_light_samples - 1) / C.initial_light_samples;
            uint32_t offset = min((uint32_t)(rnd(sg) * stride), stride - 1);
            const v3 n = gp.n;
            const rtarget::Ctx ctx = rtarget::make_ctx(n, gp.rd, gp.brdf);
            const float ratio = (float)C.initial_brdf_samples / (float)(C.initial_light_samples + C.initial_brdf_samples);
            Ris s = empty_ris();
            auto candidate = [&](v3 ld, v3 ldir, float lpdf, float llum) {
                float targetPdf = fmaxf(0.f, llum * rtarget::eval_brdf(ctx, ldir));   // rtarget::target with the precomputed luminance
                float sourcePdf = lerpf(lpdf, rtarget::pdf_brdf(ctx, ldir), ratio);  // res.slang:79-91
                float w = mr_div(targetPdf, sourcePdf);                                // res.slang:93-113
                s.weightSum += w; s.M += 1.f;
                if (rnd(sg) * s.weightSum < w) { s.light_data = ld; s.inv_pdf = lpdf; s.weight = targetPdf; s.lum = llum; }
            };
            {
                // light_tile_size is 1024 in every configuration of the reference: the wrap is a mask then (a 32-bit modulo by a run-time value costs ~35 instructions, 32
                // times per pixel and sample). Requesting several candidate records ahead of their use was measured too: no gain (profiles/r04_ab_igen_ahead.txt) — the
                // loop is bound by its ~350 instructions per candidate (three IEEE divisions, a normalisation, the GGX pdf), not by its gathers.
                const uint32_t ts_mask = (uint32_t)C.light_tile_size - 1u; const bool ts_pow2 = ((uint32_t)C.light_tile_size & ts_mask) == 0u;
                for (uint32_t i = 0; i < (uint32_t)C.initial_light_samples; ++i) {
                    const uint32_t v = offset + i * stride;
                    const uint32_t index = tileOffset + (ts_pow2 ? (v & ts_mask) : v % (uint32_t)C.light_tile_size);
                    if (COMPACT) {
                        const float4 cx = tile_aux[index];
                        candidate(V3(cx.w != 0.f ? 1.f : 0.f, cx.x, cx.y), oct_decode(V2(cx.x, cx.y)), cx.w, cx.z);
                    } else {
                        const float4 ax = tile_aux[2 * (size_t)index], bx = tile_aux[2 * (size_t)index + 1];
                        candidate(V3(bx.y, bx.z, bx.w), V3(ax.x, ax.y, ax.z), bx.x, ax.w);
                    }
                }
            }
            for (int i = 0; i < C.initial_brdf_samples; ++i) {
                float xa = rnd(sg), xb = rnd(sg), xc = rnd(sg);
                v3 ld = V3(0.f); float lpdf = 0.f; v3 dir;
                if (rtarget::sample_brdf(ctx, xa, xb, xc, dir)) {
                    lpdf = pdf_li(E, dir);
                    v2 o = oct_encode(dir);
                    ld = V3(1.0f, o.x, o.y);
                }
                if (ld.x < 0.1f) { s.M += 1.f; continue; }
                const float elum = luminance(env_radiance(E, dir));
                float targetPdf = target_lum(ctx, elum, dir);
                float sourcePdf = lerpf(lpdf, rtarget::pdf_brdf(ctx, dir), ratio);
                float w = mr_div(targetPdf, sourcePdf);
                s.weightSum += w; s.M += 1.f;
                // the carried luminance is that of the direction later passes will SEE — the octahedral code decoded again —, not of `dir` itself
                if (rnd(sg) * s.weightSum < w) { s.light_data = ld; s.inv_pdf = lpdf; s.weight = targetPdf; s.lum = luminance(env_radiance(E, oct_decode(V2(ld.y, ld.z)))); }
            }
            if (s.light_data.x > 0.1f) { want = true; rpos = load_gpos(G, pi); rdir = oct_decode(V2(s.light_data.y, s.light_data.z)); }
            // reservoir as if the sample is visible; k_initial_resolve empties it when the shadow ray hits (:269-281)
            s.weight = s.weight > 0.f ? mr_div(mr_div(s.weightSum, s.M), s.weight) : 0.f;
            s.M = 1.f;
            store_ris(R, sv, s);
        }
    }
    uint32_t slot = block_append(q_count, want);
    if (want) put_ray(q, slot, rpos, rdir, C.vis_near);
    if (sv < NV) slot_out[sv] = want ? (int32_t)slot : -1;
}
__global__ void __launch_bounds__(MR_BLOCK) k_initial_resolve(ResD R, int N, const int32_t* __restrict__ slot, const int32_t* __restrict__ hit) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= N) return;
    int s = slot[pi];
    if (s >= 0 && hit[s]) store_res(R, pi, V3(0.f), 0.f, 1, 0.f);  // createEmpty, then M := 1
    else if (s >= 0 && R.rec) R.rec[2 * (size_t)pi + 1].z = __int_as_float(1);   // the candidate's shadow ray came back free: visible from this pixel
}

// ---------------------------------------------------------------- temporal resampling (TemporalResampling.slang:23-135)
// the merge itself (TemporalResampling.slang:60-135) for one pixel: current reservoir `cur`, history `prev` taken from pixel q (G-buffer entry gq; `same` = q is this pixel).
// Returns false when the neighbour is rejected (the current reservoir stays as it is); sg = the pixel's generator after the two jitter draws.
MR_DEV bool temporal_merge(const mirres_config_t& C, const EnvD& E, const GPix& gc, const GPix& gq, ResV cur, ResV prev, bool same, uint32_t& sg, Ris& s) {
    const v3 n = gc.n; const float depth = gc.depth;
    const v3 pn = gq.n; const float pdepth = gq.depth;
    prev.M = min(prev.M, cur.M * C.max_history);
    if (!(dot(n, pn) >= 0.5f && fabsf(depth - pdepth) <= 0.1f * depth)) return false;  // isValidNeighbor res.slang:63-68
    const rtarget::Ctx ctx = rtarget::make_ctx(n, gc.rd, gc.brdf);
    const rtarget::Ctx pctx = rtarget::make_ctx(pn, gq.rd, gq.brdf);
    s = empty_ris();
    v3 ldir = oct_decode(V2(cur.light_data.y, cur.light_data.z));
    const float clum = sample_lum(E, cur, ldir);
    float targetPdf = target_lum(ctx, clum, ldir);
    {
        float w = targetPdf * cur.weight * cur.M;  // res.slang:116-134
        s.weightSum += w; s.M += cur.M;
        if (rnd(sg) * s.weightSum < w) { s.light_data = cur.light_data; s.inv_pdf = cur.light_pdf; s.weight = targetPdf; s.vcode = cur.vcode; s.lum = clum; }
    }
    v3 pldir = oct_decode(V2(prev.light_data.y, prev.light_data.z));
    const float plum = sample_lum(E, prev, pldir);
    float preTarget = target_lum(ctx, plum, pldir);
    bool usedPrev;
    {
        float w = preTarget * prev.weight * prev.M;
        s.weightSum += w; s.M += prev.M;
        usedPrev = rnd(sg) * s.weightSum < w;
        if (usedPrev) { s.light_data = prev.light_data; s.inv_pdf = prev.light_pdf; s.weight = preTarget; s.vcode = same ? prev.vcode : 0; s.lum = plum; }
    }
    v3 sdir = oct_decode(V2(s.light_data.y, s.light_data.z));
    // the selected sample is the current one, the history's, or none (nothing selected: s.weight = 0 and the result is the empty reservoir whatever this luminance is)
    const float slum = cur.has_lum ? s.lum : luminance(env_radiance(E, sdir));
    float currentPdf = target_lum(ctx, slum, sdir);
    float prevPdf = target_lum(pctx, slum, sdir);
    float normalization = mr_div(usedPrev ? prevPdf : currentPdf, cur.M * currentPdf + prev.M * prevPdf);
    s.weight = s.weight > 0.f ? mr_div(s.weightSum * normalization, s.weight) : 0.f;
    return true;
}
// the history pixel of (x, y): the jitter int2(pixel + rnd) (TemporalResampling.slang:40-52; motion vectors optional) — almost always the pixel itself, but
// (float)x + u rounds up to x + 1 for the largest u. Returns false when it leaves the frame.
MR_DEV bool temporal_history_pixel(uint32_t x, uint32_t y, float mvx, float mvy, int fx, int fy, uint32_t& sg, int& ppx, int& ppy) {
    float jx = rnd(sg), jy = rnd(sg);
    ppx = (int)(((float)x + mvx * (float)(uint32_t)fx) + (jx * 1.f - 0.f));
    ppy = (int)(((float)y + mvy * (float)(uint32_t)fy) + (jy * 1.f - 0.f));
    return !(ppx >= fx || ppx < 0 || ppy >= fy || ppy < 0);
}
__global__ void __launch_bounds__(MR_BLOCK) k_temporal(mirres_config_t C, EnvD E, GBufD G, GBufD P, ResD R, ResD PR, const float* __restrict__ motion,
                                                       uint32_t frameIndex, int fx, int fy, int N, int y_off) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= N) return;
    const GPix gc = load_gpix(G, pi);
    if (gc.occ < 0.1f) return;
    const uint32_t x = (uint32_t)(pi % fx), y = (uint32_t)(pi / fx);
    uint32_t sg = seed_generator(x, y + (uint32_t)y_off, frameIndex);
    float mvx = motion ? motion[2 * (size_t)pi] : 0.f, mvy = motion ? motion[2 * (size_t)pi + 1] : 0.f;
    int ppx, ppy;
    if (!temporal_history_pixel(x, y, mvx, mvy, fx, fy, sg, ppx, ppy)) return;
    const size_t qi = (size_t)ppy * fx + ppx;
    const GPix gq = load_gpix(P, qi);
    if (gq.occ < 0.1f) return;
    Ris s;
    if (temporal_merge(C, E, gc, gq, load_res(R, pi), load_res(PR, qi), qi == (size_t)pi, sg, s)) store_ris(R, pi, s);
}

// ---------------------------------------------------------------- spatial resampling (SpatialResampling.slang:178-322)
MR_DEV float m_factor(float q0, float q1) { return q0 == 0.f ? 1.f : clampf(mrf_pow2k(fminf(mr_div(q1, q0), 1.f), 3), 0.f, 1.f); }
MR_DEV float pairwise_mis(float q0, float q1, float N0, float N1) { return (q1 == 0.f) ? 0.f : mr_div(N0 * q0, q0 * N0 + q1 * N1); }

// neighbour acceptance in the reference's order of `continue`s (:236-258): in bounds -> normal / depth similar -> neighbour reservoir M != 0 ->
// neighbour is foreground. The loads of all candidate neighbours are issued before any test is looked at (a runtime loop with early-outs made
// each of the ~15 gathers of a pixel wait for the previous one: the kernel spent its time on dependent L2 round trips); the tests are unchanged.
// MR_SGEN_PX pixels per thread (rows of a MR_SGEN_TILE x MR_SGEN_TILE tile, MR_SGEN_TILE^2 / MR_SGEN_PX threads per block): the block still reserves its rays with ONE
// queue atomic, i.e. MR_SGEN_PX times fewer atomics per launch on the one head word (a word takes ~88 returning atomics per microsecond; a 1600 x 1600 frame in
// 256-pixel blocks is 10 000 of them in a 235 us kernel).
#ifndef MR_SGEN_PX
#define MR_SGEN_PX 1
#endif
MR_DEV int tile_pixel_v(int fx, int fy, int tw, int N, int vt) {   // tile_pixel for a virtual thread index vt in [0, tw * tw)
    const int tiles_x = (fx + tw - 1) / tw, tiles_y = (fy + tw - 1) / tw;
    int tx, ty;
#if MR_TILE_MAP == 0
    tx = blockIdx.x % tiles_x; ty = blockIdx.x / tiles_x;
#elif MR_TILE_MAP == 1
    const int t = (int)(blockIdx.x & 7u) * ((tiles_x * tiles_y + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (t >= tiles_x * tiles_y) return N;
    tx = t % tiles_x; ty = t / tiles_x;
#else
    const int ch = MR_CHUNK_PX / tw, chunks_x = (tiles_x + ch - 1) / ch;
    const int j = (int)(blockIdx.x >> 3), chunk = (int)(blockIdx.x & 7u) + 8 * (j / (ch * ch)), w = j % (ch * ch);
    tx = (chunk % chunks_x) * ch + w % ch; ty = (chunk / chunks_x) * ch + w / ch;
    if (tx >= tiles_x || ty >= tiles_y) return N;
#endif
    const int x = tx * tw + (vt % tw), y = ty * tw + (vt / tw);
    return (x < fx && y < fy) ? y * fx + x : N;   // N = "no pixel"
}
// ITEMS (mirres_render's chain: packed pixel records and reservoirs exist): the queue receives one (canonical pixel, neighbour pixel) pair per accepted
// neighbour — 8 bytes instead of two 32-byte rays — and k_trace_any4q<.., SRC = 1> forms the rays (engine.hpp RaySrc). Forming and writing the rays was 135 of
// this kernel's 214 us per sample (five position / light gathers per pixel, 290 MB of ray records per launch).
struct RowSet { int a, b, mode; };    // mode 0: every row; 1: rows in [a, b); 2: rows outside [a, b)  (the interior / border parts of a strip's spatial pass)
MR_DEV bool row_in(const RowSet& r, int pi, int fx) { if (r.mode == 0) return true; const int y = pi / fx; const bool in = y >= r.a && y < r.b; return r.mode == 1 ? in : !in; }
template <int MR_MAX_NB, bool ITEMS = false>   // 5 (the reference's neighbour count) or 8: bounds the unrolled gathers, i.e. the registers held
__global__ void __launch_bounds__(MR_SGEN_BLOCK / MR_SGEN_PX) k_spatial_gen(mirres_config_t C, GBufD G, ResD PR, const float* __restrict__ noff, uint32_t frameIndex,
                                                          int fx, int fy, int N, int y_off, const float* __restrict__ occ_own, Ray* __restrict__ q,
                                                          uint32_t* __restrict__ q_count, int32_t* __restrict__ slot_out, uint32_t* __restrict__ mask_out, RowSet rows, unsigned long long* __restrict__ mark_dead,
                                                          int band_y0, int band_rows) {      // the launch covers rows [band_y0, band_y0 + band_rows) (whole frame: 0, fy)
    // strip sharding: y_off = global row of local row 0 (seeds), occ_own = occupancy with the halo rows zeroed (which pixels this rank merges);
    // neighbours are tested against the true G-buffer, halo rows included
    const int k = min(C.neighbor_count, MR_MAX_NB);
    int pis[MR_SGEN_PX]; uint32_t masks[MR_SGEN_PX]; int nbs[MR_SGEN_PX][MR_MAX_NB]; uint32_t cnt_all = 0;
#pragma unroll
    for (int px = 0; px < MR_SGEN_PX; px++) {
    int pi = tile_pixel_v(fx, band_rows, MR_SGEN_TILE, fx * band_rows, (int)threadIdx.x + px * (MR_SGEN_BLOCK / MR_SGEN_PX));
    pi = pi < fx * band_rows ? pi + band_y0 * fx : N;
    if (pi < N && !row_in(rows, pi, fx)) pi = N;       // not this launch's rows: no pixel (nothing read, nothing written)
    uint32_t mask = 0, cnt = 0;
    int nb[MR_MAX_NB];
#pragma unroll
    for (int i = 0; i < MR_MAX_NB; i++) nb[i] = -1;
    GPix gc; gc.occ = 0.f;
    if (pi < N) { gc = load_gpix(G, pi); if (occ_own) gc.occ = occ_own[pi]; }
    if (pi < N && !(gc.occ < 0.1f)) {
        const int x = pi % fx, y = pi / fx;
        uint32_t sg = seed_generator((uint32_t)x, (uint32_t)(y + y_off), frameIndex);
        const uint32_t startIndex = (uint32_t)(rnd(sg) * C.neighbor_offset_count);
        const v3 n = gc.n; const float depth = gc.depth;
        float4 nd[MR_MAX_NB]; float nocc[MR_MAX_NB]; int nM[MR_MAX_NB];
#pragma unroll
        for (int i = 0; i < MR_MAX_NB; i++) {
            if (i < k) {
                const uint32_t ni = (startIndex + (uint32_t)i) & (uint32_t)(C.neighbor_offset_count - 1);
                const int nx = x + (int)(noff[2 * ni] * C.gather_radius), ny = y + (int)(noff[2 * ni + 1] * C.gather_radius);
                if (nx >= 0 && ny >= 0 && nx < fx && ny < fy) nb[i] = ny * fx + nx;
            }
        }
#pragma unroll
        for (int i = 0; i < MR_MAX_NB; i++) {
            if (nb[i] >= 0) {
                const size_t qi = (size_t)nb[i];
                if (G.rec) { nd[i] = G.rec[4 * qi]; nocc[i] = G.rec[4 * qi + 1].w; }
                else {
                    nd[i] = reinterpret_cast<const float4*>(G.normal_depth)[qi];
                    // packed reservoirs exist only inside mirres_render, where the input reservoirs were written by this frame's initial /
                    // temporal passes over this very G-buffer: M != 0 there already means "foreground" (background pixels store M = 0), so the
                    // occupancy gather — one of the three cache lines a candidate neighbour costs — is redundant
                    nocc[i] = PR.rec ? 1.0f : G.occ[qi];
                }
                nM[i] = res_M(PR, qi);
            }
        }
#pragma unroll
        for (int i = 0; i < MR_MAX_NB; i++) {
            bool ok = nb[i] >= 0;
            if (ok) {
                const v3 nn = V3(nd[i].x, nd[i].y, nd[i].z);
                ok = dot(n, nn) >= 0.5f && fabsf(depth - nd[i].w) <= 0.1f * depth;   // isValidNeighbor (res.slang:63-68)
                ok = ok && nM[i] != 0 && !(nocc[i] < 0.1f);
            }
            if (ok) { mask |= 1u << i; cnt++; } else nb[i] = -1;
        }
    }
    pis[px] = pi; masks[px] = mask; cnt_all += cnt;
#pragma unroll
    for (int i = 0; i < MR_MAX_NB; i++) nbs[px][i] = nb[i];
    }
    uint32_t s = block_append(q_count, cnt_all > 0, 2 * cnt_all);
#pragma unroll
    for (int px = 0; px < MR_SGEN_PX; px++) {
        const int pi = pis[px]; const uint32_t mask = masks[px];
        if (ITEMS) {
            if (mask) {
                slot_out[pi] = (int32_t)s;
                uint2* const qi = reinterpret_cast<uint2*>(q) + (s >> 1);     // s is even: one pair per two rays
                int j = 0;
#pragma unroll
                for (int i = 0; i < MR_MAX_NB; i++) {
                    if (!(mask & (1u << i))) continue;
                    // pair (a, b) = rays s + 2j: pixel a towards pixel b's light (canonical towards the neighbour's), and s + 2j + 1: b towards a's light
                    qi[j++] = make_uint2((uint32_t)pi, (uint32_t)nbs[px][i]);
                }
                s += 2 * (uint32_t)j;
            } else if (pi < N) slot_out[pi] = -1;
            if (pi < N) mask_out[pi] = mask;
            continue;
        }
        if (mask) {
            const v3 cpos = load_gpos(G, pi); const v3 cl = res_light(PR, pi); const v3 cdir = oct_decode(V2(cl.y, cl.z));
            // mark_dead (the counting mode of mirres_render's chain): the rays the production queue does not trace (engine.hpp RaySrc::skip_dead: the light reservoir
            // carries luminance 0) keep their slots but get an empty interval — the counted kernel then charges them one root-box test, as the production kernel does
            const bool c_dead = mark_dead && res_lum(PR, pi) == 0.f;
            v3 nl[MR_MAX_NB], np[MR_MAX_NB]; bool n_dead[MR_MAX_NB];
#pragma unroll
            for (int i = 0; i < MR_MAX_NB; i++) if (mask & (1u << i)) { nl[i] = res_light(PR, (size_t)nbs[px][i]); np[i] = load_gpos(G, (size_t)nbs[px][i]);
                                                                       n_dead[i] = mark_dead && res_lum(PR, (size_t)nbs[px][i]) == 0.f; }
            if (pi < N) slot_out[pi] = (int32_t)s;
#pragma unroll
            for (int i = 0; i < MR_MAX_NB; i++) {
                if (!(mask & (1u << i))) continue;
                const v3 ndir = oct_decode(V2(nl[i].y, nl[i].z));
                const bool da = n_dead[i], db = c_dead;
                put_ray(q, s, cpos, ndir, C.vis_near, da ? -1.f : 1e7f);        // canonical pixel towards the neighbour's light
                put_ray(q, s + 1, np[i], cdir, C.vis_near, db ? -1.f : 1e7f);   // neighbour towards the canonical light
                s += 2;
                if (mark_dead && (da || db)) atomicAdd(mark_dead, (unsigned long long)((int)da + (int)db));
            }
        } else if (pi < N) slot_out[pi] = -1;
        if (pi < N) mask_out[pi] = mask;
    }
}

#ifndef MR_SRES_HIT2
#define MR_SRES_HIT2 1
#endif
#ifndef MR_SRES_WAVES
#define MR_SRES_WAVES 0      // experiment: waves per SIMD the register allocator must make room for (0 = its own choice: 140 VGPRs, three waves)
#endif
#if MR_SRES_WAVES
#define MR_SRES_ATTR __attribute__((amdgpu_waves_per_eu(MR_SRES_WAVES, MR_SRES_WAVES)))
#else
#define MR_SRES_ATTR
#endif
// the spatial merge of ONE pixel (SpatialResampling.slang:200-322) from the generator's acceptance mask and the traced hit bits; false: the pixel is background (or,
// under strip sharding, not this rank's): its output reservoir is empty
template <int MR_MAX_NB>
MR_DEV bool spatial_pixel(const mirres_config_t& C, const EnvD& E, const GBufD& G, const ResD& PR, const float* __restrict__ noff, uint32_t frameIndex, int fx, int y_off,
                          const float* __restrict__ occ_own, const int32_t* __restrict__ slot, const uint32_t* __restrict__ mask_in, const int32_t* __restrict__ hit,
                          int pi, Ris& s, GPix* gc_out = nullptr) {
    const GPix gc = load_gpix(G, pi);
    if (gc_out) *gc_out = gc;          // (the fused temporal merge needs the same record: a second load_gpix after the reservoir store could not be merged with this one)
    if ((occ_own ? occ_own[pi] : gc.occ) < 0.1f) return false;
    const int x = pi % fx, y = pi / fx;
    uint32_t sg = seed_generator((uint32_t)x, (uint32_t)(y + y_off), frameIndex);
    const v3 n = gc.n;
    const rtarget::Ctx ctx = rtarget::make_ctx(n, gc.rd, gc.brdf);
    s = empty_ris();
    const uint32_t startIndex = (uint32_t)(rnd(sg) * C.neighbor_offset_count);
    ResV cur = load_res(PR, pi);
    const v3 cdir = oct_decode(V2(cur.light_data.y, cur.light_data.z));
    const float clum = sample_lum(E, cur, cdir);
    const float curTarget = target_lum(ctx, clum, cdir);
    s.canonicalWeight = 1.f;
    uint32_t validNeighbors = 1;
    const uint32_t k = (uint32_t)C.neighbor_count;
    const uint32_t mask = mask_in[pi];
    int hs = slot[pi];
    // the accepted neighbours' pixel data and reservoirs are fetched up front (independent gathers in flight together), then merged in order
    GPix gnb[MR_MAX_NB]; ResV rnb[MR_MAX_NB];
#pragma unroll
    for (int i = 0; i < MR_MAX_NB; i++) {
        if ((uint32_t)i < k && (mask & (1u << i))) {
            const uint32_t ni = (startIndex + (uint32_t)i) & (uint32_t)(C.neighbor_offset_count - 1);
            const int nx = x + (int)(noff[2 * ni] * C.gather_radius), ny = y + (int)(noff[2 * ni + 1] * C.gather_radius);
            const size_t qi = (size_t)ny * fx + nx;
            gnb[i] = load_gpix(G, qi); rnb[i] = load_res(PR, qi);
        }
    }
#pragma unroll
    for (int i = 0; i < MR_MAX_NB; i++) {
        if (!((uint32_t)i < k && (mask & (1u << i)))) continue;
        const v3 nn = gnb[i].n;
        const ResV nbr = rnb[i];
        const rtarget::Ctx nctx = rtarget::make_ctx(nn, gnb[i].rd, gnb[i].brdf);
        ++validNeighbors;
        const v3 ndir = oct_decode(V2(nbr.light_data.y, nbr.light_data.z));
        const float nlum = sample_lum(E, nbr, ndir);
#if MR_SRES_HIT2   // the two answers of a pair are neighbours in the hit array and a pixel's first slot is even: one 8-byte load instead of two 4-byte ones
        const int2 hh = reinterpret_cast<const int2*>(hit)[hs >> 1];
        const float canonicalVis = hh.x ? 0.f : 1.f, candidateVis = hh.y ? 0.f : 1.f;
#else
        const float canonicalVis = hit[hs] ? 0.f : 1.f, candidateVis = hit[hs + 1] ? 0.f : 1.f;
#endif
        hs += 2;
        // streamingResampleStepMisUnbiased (res.slang:173-213)
        float candTarget = target_lum(nctx, nlum, ndir);
        float candAtOther = target_lum(ctx, nlum, ndir);
        float canonAtOther = target_lum(nctx, clum, cdir);
        candAtOther *= canonicalVis;
        canonAtOther *= candidateVis;
        float N0 = (float)((uint32_t)nbr.M * k), N1 = (float)cur.M;
        float m0 = pairwise_mis(candTarget, candAtOther, N0, N1);
        float m1 = 1.f - pairwise_mis(canonAtOther, curTarget, N0, N1);
        float w = candAtOther * nbr.weight * m0;
        // state.M (+= M_j * min(mFactor..)) is overwritten with M_canonical below (:302), so it is not tracked
        s.weightSum += w;
        s.canonicalWeight += m1;
        if (rnd(sg) * s.weightSum < w) { s.light_data = nbr.light_data; s.inv_pdf = nbr.light_pdf; s.weight = candAtOther; s.vcode = canonicalVis > 0.f ? 1 : 2; s.lum = nlum; }
    }
    {   // streamingResampleFinalizeMis (res.slang:215-232)
        float w = curTarget * cur.weight * s.canonicalWeight;
        s.weightSum += w;
        if (rnd(sg) * s.weightSum < w) { s.light_data = cur.light_data; s.inv_pdf = cur.light_pdf; s.weight = curTarget; s.vcode = cur.vcode; s.lum = clum; }
    }
    s.M = (float)cur.M;
    s.weight = s.weight > 0.f ? mr_div(mr_div(s.weightSum, (float)validNeighbors), s.weight) : 0.f;
    return true;
}
// what store_ris leaves in memory for a merge result, as the next pass would load it (packed records)
MR_DEV ResV stored_res(const Ris& s, bool fg) {
    ResV r; r.has_lum = true;
    if (!fg || isinf(s.weight) || isnan(s.weight)) { r.light_data = V3(0.f); r.light_pdf = 0.f; r.M = 0; r.weight = 0.f; r.vcode = 0; r.lum = 0.f; return r; }
    r.light_data = s.light_data; r.light_pdf = s.inv_pdf; r.M = (int)s.M; r.weight = s.weight; r.vcode = s.vcode; r.lum = s.lum;
    return r;
}
// FUSE (mirres_render's chain, round 4): the temporal merge of the NEXT sample (TemporalResampling.slang:23-135) runs in the same thread right after this sample's
// spatial merge. Its history is the spatial output of the pixel the jitter selects — the pixel itself (in registers: no reservoir round trip, no G-buffer
// re-read, no launch) or, when (float)x + u rounds up (a few hundred pixels of a 1600 x 1600 frame per sample), its right / lower neighbour, whose spatial merge this
// thread then recomputes from the same inputs (the neighbour's own thread may not have stored it yet). NR = the next sample's initial reservoirs, merged in place.
template <int MR_MAX_NB, bool FUSE>
__global__ void MR_SRES_ATTR __launch_bounds__(MR_SRES_TILE * MR_SRES_TILE) k_spatial_resolve(mirres_config_t C, EnvD E, GBufD G, ResD R, ResD PR, const float* __restrict__ noff,
                                                              uint32_t frameIndex, int fx, int fy, int N, int y_off, const float* __restrict__ occ_own,
                                                              const int32_t* __restrict__ slot, const uint32_t* __restrict__ mask_in, const int32_t* __restrict__ hit,
                                                              uint32_t* __restrict__ reset_counter, uint32_t* __restrict__ reset_heads, ResD NR, uint32_t next_frame, RowSet rows,
                                                              int band_y0, int band_rows) {
    // mirres_render's chain: the shadow-ray launch of this pass has finished (stream order), so the ray counter and the traversal work heads it used
    // are zeroed here for the next sample's pass instead of by two separate fill launches per sample
    if (reset_heads && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < MR_WSET; i += MR_SRES_TILE * MR_SRES_TILE) reset_heads[i] = 0u;
        if (threadIdx.x == 0) *reset_counter = 0u;
    }
    int pi = tile_pixel(fx, band_rows, MR_SRES_TILE, fx * band_rows);
    pi = pi < fx * band_rows ? pi + band_y0 * fx : N;
    if (pi >= N || !row_in(rows, pi, fx)) return;
    Ris s; GPix gc;
    const bool fg = spatial_pixel<MR_MAX_NB>(C, E, G, PR, noff, frameIndex, fx, y_off, occ_own, slot, mask_in, hit, pi, s, &gc);
    if (!fg) { store_zero(R, pi); return; }
    store_ris(R, pi, s);
    if (FUSE) {
        const uint32_t x = (uint32_t)(pi % fx), y = (uint32_t)(pi / fx);
        uint32_t sg = seed_generator(x, y + (uint32_t)y_off, next_frame);
        int ppx, ppy;
        if (!temporal_history_pixel(x, y, 0.f, 0.f, fx, fy, sg, ppx, ppy)) return;
        const int qi = ppy * fx + ppx;
        if (occ_own) gc.occ = occ_own[pi];
        GPix gq = gc; ResV prev;
        if (qi == pi) prev = stored_res(s, true);
        else {
            gq = load_gpix(G, qi); if (occ_own) gq.occ = occ_own[qi];
            if (gq.occ < 0.1f) return;
            Ris sq;
            const bool qfg = spatial_pixel<MR_MAX_NB>(C, E, G, PR, noff, frameIndex, fx, y_off, occ_own, slot, mask_in, hit, qi, sq);
            prev = stored_res(sq, qfg);
        }
        Ris t;
        if (temporal_merge(C, E, gc, gq, load_res(NR, pi), prev, qi == pi, sg, t)) store_ris(NR, pi, t);
    }
}

// ---------------------------------------------------------------- final-sample visibility + evaluation (EvaluateFinalSamples.slang:84-188)
__global__ void __launch_bounds__(MR_GEN_BLOCK) k_vis_gen(float vis_near, const float* __restrict__ pos, ResD R, int N, int NV, Ray* __restrict__ q,
                                                      uint32_t* __restrict__ q_count, int32_t* __restrict__ slot_out) {
    const int sv = blockIdx.x * blockDim.x + threadIdx.x;   // slot k * N + pixel (reservoir per slot, position per pixel)
    bool want = false; v3 rp = V3(0.f), rd = V3(0.f);
    if (sv < NV) {
        v3 ld = res_light(R, sv);
        if (ld.x > 0.1f && res_vcode(R, sv) == 0) { want = true; rp = ld3(pos, sv % N); rd = oct_decode(V2(ld.y, ld.z)); }
    }
    uint32_t slot = block_append(q_count, want);
    if (want) put_ray(q, slot, rp, rd, vis_near);
    if (sv < NV) slot_out[sv] = want ? (int32_t)slot : -1;
}
// Final visibility + EvaluateFinalSamples + FinalShading of the K samples of a batch, one thread per pixel, samples in ascending order: the three
// per-sample kernels (k_vis_resolve, k_eval_final, k_final_shading<ACC>) back to back on registers — same arithmetic, same summation order.
__global__ void __launch_bounds__(MR_BLOCK) k_final_direct(EnvD E, const float* __restrict__ occ, const float* __restrict__ normal, const float* __restrict__ ray_dir,
                                                           const float* __restrict__ kd, const float* __restrict__ rm, ResD R, const int32_t* __restrict__ slot,
                                                           const int32_t* __restrict__ hit, int N, int K, float* __restrict__ color, float* __restrict__ diff_light,
                                                           float* __restrict__ spec_light, float4* __restrict__ tape) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= N) return;
    const v3 n = ld3(normal, pi), rd = ld3(ray_dir, pi), diffuse = ld3(kd, pi);
    const float rough = rm[2 * (size_t)pi], metallic = rm[2 * (size_t)pi + 1];
    const bool fg = occ[pi] > 0.1f;
    v3 ac = ld3(color, pi), ad = ld3(diff_light, pi), as = ld3(spec_light, pi);
    const v3 bg = fg ? V3(0.f) : env_le(ngp_dir(rd), E.tex, E.W, E.H);
    for (int k = 0; k < K; k++) {
        const size_t sv = (size_t)k * N + pi;
        v3 c = V3(0.f), ldiff = V3(0.f), lspec = V3(0.f);
        if (tape) {   // what the backward needs of this sample: the merged reservoir and its visibility (mirres_render_bwd)
            const ResV rv = load_res(R, sv);
            const int sl0 = slot[sv];
            float4 a, b; a.x = rv.light_data.x; a.y = rv.light_data.y; a.z = rv.light_data.z; a.w = rv.light_pdf;
            b.x = __int_as_float(rv.M); b.y = rv.weight; b.z = rv.vcode ? (rv.vcode == 1 ? 1.0f : 0.0f) : ((sl0 >= 0 && hit[sl0]) ? 0.0f : 1.0f); b.w = 0.f;
            tape[2 * sv] = a; tape[2 * sv + 1] = b;
        }
        if (fg) {
            const int sl = slot[sv], vc = res_vcode(R, sv);
            const float vis = vc ? (vc == 1 ? 1.0f : 0.0f) : ((sl >= 0 && hit[sl]) ? 0.0f : 1.0f);   // k_vis_resolve, or the earlier stage's answer for the same ray
            const v3 ld = res_light(R, sv);
            v3 dir = V3(0.f), Li = V3(0.f); float dist = 0.f;
            if (ld.x > 0.1f) {                                                           // k_eval_final
                const v3 ldir = oct_decode(V2(ld.y, ld.z));
                const v3 em = env_radiance(E, ldir);
                if (vis > 0.f) { dir = ldir; dist = 1e6f; Li = res_weight(R, sv) * em; }
            }
            v3 dv = V3(0.f), sv3 = V3(0.f);                                              // k_final_shading
            if (dist > 0.f) {
                shade::Frame fr = shade::create_frame(n);
                v3 wi = shade::to_local(fr, -rd), wo = shade::to_local(fr, dir);
                shade::Lobes L = shade::lobes(diffuse, rough, metallic, rd, n);
                if (L.pD > 0.f) dv = shade::diffuse_light(wi, wo) * Li;
                if (L.pS > 0.f) sv3 = shade::specular_eval(wi, wo, L.specular, L.alpha) * Li;
            }
            c = diffuse * (1.0f - metallic) * dv + sv3;
            ldiff = dv; lspec = sv3;
        } else c = bg;
        ac = ac + c; ad = ad + ldiff; as = as + lspec;
    }
    st3(color, pi, ac); st3(diff_light, pi, ad); st3(spec_light, pi, as);
}
__global__ void __launch_bounds__(MR_BLOCK) k_vis_resolve(int N, const int32_t* __restrict__ slot, const int32_t* __restrict__ hit, float* __restrict__ vis) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= N) return;
    int s = slot[pi];
    vis[pi] = (s >= 0 && hit[s]) ? 0.0f : 1.0f;
}
__global__ void __launch_bounds__(MR_BLOCK) k_eval_final(EnvD E, ResD R, const float* __restrict__ vis, int N, float* __restrict__ fdir,
                                                         float* __restrict__ fdist, float* __restrict__ fLi) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= N) return;
    ResV cur = load_res(R, pi);
    v3 od = V3(0.f), Li = V3(0.f); float dist = 0.f;
    if (cur.light_data.x > 0.1f) {
        v3 ldir = oct_decode(V2(cur.light_data.y, cur.light_data.z));
        v3 em = env_radiance(E, ldir);
        if (vis[pi] > 0.f) { od = ldir; dist = 1e6f; Li = cur.weight * em; }
    }
    st3(fdir, pi, od); fdist[pi] = dist; st3(fLi, pi, Li);
}
// backward of k_eval_final w.r.t. the environment texels: Li = W * bilinear(env, ngp_dir(ldir))  (EvaluateFinalSamples.slang:129-188 .bwd)
__global__ void __launch_bounds__(MR_BLOCK) k_eval_final_bwd(EnvD E, ResD R, const float* __restrict__ vis, int N, const float* __restrict__ gLi,
                                                             float* __restrict__ genv) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= N) return;
    ResV cur = load_res(R, pi);
    if (!(cur.light_data.x > 0.1f) || !(vis[pi] > 0.f)) return;
    v3 ldir = oct_decode(V2(cur.light_data.y, cur.light_data.z));
    int idx[4]; float w[4];
    if (!env_le_footprint(ngp_dir(ldir), E.W, E.H, idx, w)) return;
    v3 g = ld3(gLi, pi) * cur.weight;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        atomicAdd(&genv[3 * (size_t)idx[k]], g.x * w[k]); atomicAdd(&genv[3 * (size_t)idx[k] + 1], g.y * w[k]); atomicAdd(&genv[3 * (size_t)idx[k] + 2], g.z * w[k]);
    }
}

// internal convention (mirres_render only; the ABI entry points never see it): light_pdf == NULL -> `light_data` points to packed reservoir records
static GBufD gbufd(const mirres_gbuf_t* g) { GBufD G; G.occ = g->occ; G.pos = g->pos; G.normal_depth = g->normal_depth; G.brdf = g->brdf; G.ray_dir = g->ray_dir; G.rec = nullptr; return G; }
static ResD resd(const mirres_res_t* r) {
    ResD R; R.light_data = r->light_data; R.light_pdf = r->light_pdf; R.M = r->M; R.weight = r->weight; R.rec = nullptr;
    if (!r->light_pdf) { R.rec = reinterpret_cast<float4*>(r->light_data); R.light_data = nullptr; }
    return R;
}
static EnvD envh(const mirres_env_t* e) { EnvD E; E.tex = e->tex; E.W = e->Wc; E.H = e->Hc; E.pdf = e->pdf; E.cdf = e->cdf; E.mpdf = e->mpdf; E.mcdf = e->mcdf; return E; }

int trace_closest_queue_counted(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, HitRec* out,
                                unsigned long long* stats, hipStream_t s);

static int ev_pair(std::vector<hipEvent_t>& pool, size_t& used, hipEvent_t** a, hipEvent_t** b) {
    if (used + 2 > pool.size()) {
        size_t old = pool.size(); pool.resize(old + 512);
        for (size_t i = old; i < pool.size(); i++) MR_HIP(hipEventCreate(&pool[i]));
    }
    *a = &pool[used]; *b = &pool[used + 1]; used += 2;
    return 0;
}
int trace_any_q(mirres_ctx* ctx, mirres_bvh* bvh, const Ray* rays, const uint32_t* count, size_t cap, int32_t* hit, hipStream_t s, int lane) {
    hipEvent_t *e0 = nullptr, *e1 = nullptr;
    if (ctx->instrument & 2) { int rc = ev_pair(ctx->ev_any, ctx->ev_any_used, &e0, &e1); if (rc) return rc; MR_HIP(hipEventRecord(*e0, s)); }
    int rc = (ctx->instrument & 1) ? trace_any_queue_counted(bvh, rays, count, cap, hit, ctx->stats, s, (ctx->instrument & 4) != 0)
                                   : trace_any_queue(bvh, rays, count, cap, hit, ctx->stats, s, lane, (ctx->instrument & 2) != 0, lane == 0 && ctx->chain_reset && ctx->chain_clean);
    if (e1) MR_HIP(hipEventRecord(*e1, s));
    return rc;
}
// queue of (origin pixel, light pixel) pairs instead of rays (engine.hpp RaySrc); never in the counting mode (callers fall back to rays there).
// Only the spatial pass uses it: the initial pass's ray must follow the candidate's direction even when its reservoir is stored empty (non-finite weight), and
// one ray per pixel is a small part of that kernel anyway.
int trace_any_items_q(mirres_ctx* ctx, mirres_bvh* bvh, const Ray* queue, const RaySrc& src, const uint32_t* count, size_t cap, int32_t* hit, hipStream_t s, int lane) {
    hipEvent_t *e0 = nullptr, *e1 = nullptr;
    if (ctx->instrument & 2) { int rc = ev_pair(ctx->ev_any, ctx->ev_any_used, &e0, &e1); if (rc) return rc; MR_HIP(hipEventRecord(*e0, s)); }
    int rc = trace_any_items_queue(bvh, reinterpret_cast<const uint2*>(queue), src, count, cap, hit, ctx->stats, s, lane, (ctx->instrument & 2) != 0,
                                   lane == 0 && ctx->chain_reset && ctx->chain_clean);
    if (e1) MR_HIP(hipEventRecord(*e1, s));
    return rc;
}
static bool ray_items_allowed(const mirres_ctx* ctx) {
    static const bool force_rays = [] { const char* e = getenv("MIRRES_SPATIAL_RAYS"); return e && e[0] == '1'; }();   // A/B: 32-byte rays everywhere, as before
    return ctx->grec && !(ctx->instrument & 1) && !force_rays;
}
int trace_closest_q(mirres_ctx* ctx, mirres_bvh* bvh, const Ray* rays, const uint32_t* count, size_t cap, HitRec* out, hipStream_t s, int lane) {
    hipEvent_t *e0 = nullptr, *e1 = nullptr;
    if (ctx->instrument & 2) { int rc = ev_pair(ctx->ev_cl, ctx->ev_cl_used, &e0, &e1); if (rc) return rc; MR_HIP(hipEventRecord(*e0, s)); }
    int rc = (ctx->instrument & 1) ? trace_closest_queue_counted(bvh, rays, count, cap, out, ctx->stats, s)
                                   : trace_closest_queue(bvh, rays, count, cap, out, ctx->stats, s, lane);
    if (e1) MR_HIP(hipEventRecord(*e1, s));
    return rc;
}
int trace_any(mirres_ctx* ctx, mirres_bvh* bvh, size_t cap, hipStream_t s) { return trace_any_q(ctx, bvh, ctx->any_rays, &ctx->counters[0], cap, ctx->any_hit, s); }
int trace_closest(mirres_ctx* ctx, mirres_bvh* bvh, size_t cap, hipStream_t s) { return trace_closest_q(ctx, bvh, ctx->cl_rays, &ctx->counters[1], cap, ctx->cl_hit, s); }

// ---- K-sample batches of the history-free ReSTIR stages (mirres_render). Initial resampling needs nothing from earlier samples, and the final
// visibility / evaluation / shading of a sample feeds only the frame totals, so K samples go through each of them in one set of launches
// (K * N slots, like the path-tracing stages); only temporal + spatial reuse stay sample by sample.
int launch_initial_batch(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_res_t* res, float* tile_data,
                         float* tile_pdf, float* tile_aux, uint32_t frame0, int K, const PtQueues* q, hipStream_t s) {
    const int N = (int)ctx->N, NV = K * N;
    const int TS = ctx->cfg.light_tile_count * ctx->cfg.light_tile_size;
    MR_HIP(hipMemsetAsync(&q->counters[0], 0, sizeof(uint32_t), s));
    k_light_tiles<<<grid_for((size_t)K * TS, MR_BLOCK), MR_BLOCK, 0, s>>>(envh(env), frame0, K * TS, TS, tile_data, nullptr, tile_pdf);       // pass 0 (+1 inside)
    static const bool compact = [] { const char* e = getenv("MIRRES_TILE_COMPACT"); return !(e && e[0] == '0'); }();
    if (compact) k_tile_aux<true><<<grid_for((size_t)K * TS, MR_BLOCK), MR_BLOCK, 0, s>>>(envh(env), K * TS, tile_data, tile_pdf, reinterpret_cast<float4*>(tile_aux));
    else k_tile_aux<false><<<grid_for((size_t)K * TS, MR_BLOCK), MR_BLOCK, 0, s>>>(envh(env), K * TS, tile_data, tile_pdf, reinterpret_cast<float4*>(tile_aux));
    (compact ? k_initial_gen<true> : k_initial_gen<false>)<<<grid_for(NV, MR_IGEN_BLOCK), MR_IGEN_BLOCK, 0, s>>>(ctx->cfg, envh(env), gbufd(g), resd(res), tile_data, tile_pdf, reinterpret_cast<const float4*>(tile_aux),
                                                                       frame0 + 2, ctx->fx, N, NV, TS, ctx->y_off, q->any_rays, &q->counters[0], q->slot_a);       // pass 2
    int rc = trace_any_q(ctx, bvh, q->any_rays, &q->counters[0], (size_t)NV, q->any_hit, s, q->lane); if (rc) return rc;
    k_initial_resolve<<<grid_for(NV, MR_BLOCK), MR_BLOCK, 0, s>>>(resd(res), NV, q->slot_a, q->any_hit);
    MR_LAUNCH_CHECK("initial_batch");
    return 0;
}
int launch_final_batch(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_env_t* env, const float* occ, const float* pos, const float* normal, const float* ray_dir,
                       const float* kd, const float* rm, const mirres_res_t* res, int K, const PtQueues* q, float* color, float* diff, float* spec, float* tape, hipStream_t s) {
    const int N = (int)ctx->N, NV = K * N;
    MR_HIP(hipMemsetAsync(&q->counters[0], 0, sizeof(uint32_t), s));
    k_vis_gen<<<grid_for(NV, MR_GEN_BLOCK), MR_GEN_BLOCK, 0, s>>>(ctx->cfg.vis_near, pos, resd(res), N, NV, q->any_rays, &q->counters[0], q->slot_a);
    int rc = trace_any_q(ctx, bvh, q->any_rays, &q->counters[0], (size_t)NV, q->any_hit, s, q->lane); if (rc) return rc;
    k_final_direct<<<grid_for(N, MR_BLOCK), MR_BLOCK, 0, s>>>(envh(env), occ, normal, ray_dir, kd, rm, resd(res), q->slot_a, q->any_hit, N, K, color, diff, spec, reinterpret_cast<float4*>(tape));
    MR_LAUNCH_CHECK("final_batch");
    return 0;
}

// spatial pass; next_res != NULL (mirres_render's chain, packed reservoirs): the temporal merge of the next sample is fused into the resolve kernel (k_spatial_resolve<., true>)
int launch_spatial(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_res_t* res, const mirres_res_t* prev_res,
                   const float* neighbor_offsets, uint32_t frameIndex, hipStream_t s, const mirres_res_t* next_res, uint32_t next_frame, const SpatialBand* band) {
    if (!ctx || !bvh || !env || !g || !res || !prev_res) { set_error("mirres_restir_spatial: null"); return MIRRES_E_ARG; }
    const int N = (int)ctx->N;
    const float* noff = neighbor_offsets ? neighbor_offsets : ctx->noff;
    const bool fold = ctx->chain_reset;                         // inside mirres_render's chain (its own stream, its own work heads)
    // a unit of the band pipeline (render.hip) works on rows [y0, y1) with the queue, hit bits, per-pixel slots and work heads of its chain stream's set
    Ray* const q_rays = band ? band->set->q : ctx->any_rays; int32_t* const q_hit = band ? band->set->hit : ctx->any_hit; uint32_t* const q_count = band ? band->set->counter : &ctx->counters[0];
    int32_t* const px_slot = band ? band->set->slot : ctx->slot_a; uint32_t* const px_mask = band ? band->set->mask : ctx->mask_a;
    const int head_set = band ? band->set->head_set : 0;
    bool& clean = band ? band->set->clean : ctx->chain_clean;
    const int by0 = band ? band->y0 : 0, gen_rows = band ? band->gen_y1 - band->y0 : ctx->fy, res_rows = band ? band->y1 - band->y0 : ctx->fy;
    if (!(fold && clean)) MR_HIP(hipMemsetAsync(q_count, 0, sizeof(uint32_t), s));
    uint32_t* const rc_ = fold ? q_count : nullptr; uint32_t* const rh_ = fold ? bvh->work + (size_t)head_set * MR_WSET : nullptr;
    const bool nb5 = ctx->cfg.neighbor_count <= 5;
    const RowSet rows = {ctx->row_a, ctx->row_b, ctx->row_mode};
    // mirres_render's chain (packed pixel records + packed reservoirs, no per-ray counters wanted): the queue carries pixel pairs and the traversal kernel forms the rays
    const bool items = ray_items_allowed(ctx) && resd(prev_res).rec;
    // rays whose answer the merge cannot see (engine.hpp RaySrc::skip_dead) are not traced; MIRRES_SKIP_DEAD=0: trace them as the reference does (A/B). In the counting
    // mode of the chain (32-byte rays, own counters) they are marked instead, so that the counters describe what production traces; never for the reference-order counts
    static const int skip_dead = [] { const char* e = getenv("MIRRES_SKIP_DEAD"); return (e && e[0] == '0') ? 0 : 1; }();
    unsigned long long* const mark_dead = (skip_dead && ctx->grec && resd(prev_res).rec && (ctx->instrument & 1) && !(ctx->instrument & 4)) ? &ctx->stats[12] : nullptr;
    const dim3 sg_grid(tile_grid(ctx->fx, gen_rows, MR_SGEN_TILE)), sg_block(MR_SGEN_BLOCK / MR_SGEN_PX);
#define MR_SGEN_ARGS ctx->cfg, gbufd(g), resd(prev_res), noff, frameIndex, ctx->fx, ctx->fy, N, ctx->y_off, ctx->occ_own, q_rays, q_count, px_slot, px_mask, rows, mark_dead, by0, gen_rows
    if (nb5 && items) k_spatial_gen<5, true><<<sg_grid, sg_block, 0, s>>>(MR_SGEN_ARGS);
    else if (nb5) k_spatial_gen<5><<<sg_grid, sg_block, 0, s>>>(MR_SGEN_ARGS);
    else if (items) k_spatial_gen<8, true><<<sg_grid, sg_block, 0, s>>>(MR_SGEN_ARGS);
    else k_spatial_gen<8><<<sg_grid, sg_block, 0, s>>>(MR_SGEN_ARGS);
#undef MR_SGEN_ARGS
    int rc;
    if (items) {
        const RaySrc src = {reinterpret_cast<const float4*>(ctx->grec), resd(prev_res).rec, ctx->cfg.vis_near, skip_dead};
        // the launch is sized by what the rows can produce (two rays per accepted neighbour): a band of a small frame does not start eight thousand waves
        const size_t cap = band ? std::min(ctx->any_cap, (size_t)gen_rows * ctx->fx * (size_t)(2 * (ctx->cfg.neighbor_count > 1 ? ctx->cfg.neighbor_count : 1))) : ctx->any_cap;
        if (band) {
            hipEvent_t *e0 = nullptr, *e1 = nullptr;
            if (ctx->instrument & 2) { int rce = ev_pair(ctx->ev_any, ctx->ev_any_used, &e0, &e1); if (rce) return rce; MR_HIP(hipEventRecord(*e0, s)); }
            rc = trace_any_items_queue(bvh, reinterpret_cast<const uint2*>(q_rays), src, q_count, cap, q_hit, ctx->stats, s, 0, (ctx->instrument & 2) != 0, fold && clean, head_set);
            if (e1) MR_HIP(hipEventRecord(*e1, s));
        } else rc = trace_any_items_q(ctx, bvh, ctx->any_rays, src, &ctx->counters[0], ctx->any_cap, ctx->any_hit, s, 0);
    } else {
        if (band) { set_error("mirres_render: the band pipeline needs the pixel-pair queue"); return MIRRES_E_STATE; }
        rc = trace_any(ctx, bvh, ctx->any_cap, s);
    }
    if (rc) return rc;
    GBufD gr = gbufd(g);
    if (ctx->grec) gr.rec = reinterpret_cast<const float4*>(ctx->grec);   // mirres_render: same values, one 64-byte record per neighbour instead of three arrays
    const bool fuse = next_res && resd(next_res).rec && resd(res).rec && gr.rec;
    const ResD NR = fuse ? resd(next_res) : resd(res);
    const int rg = tile_grid(ctx->fx, res_rows, MR_SRES_TILE), rb = MR_SRES_TILE * MR_SRES_TILE;
#define MR_SRES_ARGS ctx->cfg, envh(env), gr, resd(res), resd(prev_res), noff, frameIndex, ctx->fx, ctx->fy, N, ctx->y_off, ctx->occ_own, px_slot, px_mask, q_hit, rc_, rh_, NR, next_frame, rows, by0, res_rows
    if (nb5 && fuse) k_spatial_resolve<5, true><<<rg, rb, 0, s>>>(MR_SRES_ARGS);
    else if (nb5) k_spatial_resolve<5, false><<<rg, rb, 0, s>>>(MR_SRES_ARGS);
    else if (fuse) k_spatial_resolve<8, true><<<rg, rb, 0, s>>>(MR_SRES_ARGS);
    else k_spatial_resolve<8, false><<<rg, rb, 0, s>>>(MR_SRES_ARGS);
#undef MR_SRES_ARGS
    if (fold) clean = true;
    MR_LAUNCH_CHECK("restir_spatial");
    return MIRRES_OK;
}

}  // namespace mr

using namespace mr;

extern "C" {

int mirres_ctx_create(mirres_ctx_t** out, int fx, int fy, const mirres_config_t* cfg) {
    if (!out || fx <= 0 || fy <= 0) { set_error("mirres_ctx_create: bad size"); return MIRRES_E_ARG; }
    mirres_ctx* c = new mirres_ctx();
    c->fx = fx; c->fy = fy; c->N = (size_t)fx * fy;
    if (cfg) c->cfg = *cfg; else mirres_default_config(&c->cfg);
    if (c->cfg.neighbor_count > 8 || c->cfg.neighbor_count < 0 || (c->cfg.neighbor_offset_count & (c->cfg.neighbor_offset_count - 1))) {
        set_error("mirres_ctx_create: neighbor_count must be <= 8 and neighbor_offset_count a power of two"); delete c; return MIRRES_E_ARG;
    }
    const size_t N = c->N;
    c->any_cap = N * (size_t)(2 * (c->cfg.neighbor_count > 1 ? c->cfg.neighbor_count : 1));
    c->cl_cap = N;
    MR_HIP(hipMalloc(&c->any_rays, sizeof(Ray) * c->any_cap));
    MR_HIP(hipMalloc(&c->any_hit, sizeof(int32_t) * c->any_cap));
    MR_HIP(hipMalloc(&c->cl_rays, sizeof(Ray) * c->cl_cap));
    MR_HIP(hipMalloc(&c->cl_hit, sizeof(HitRec) * c->cl_cap));
    MR_HIP(hipMalloc(&c->counters, sizeof(uint32_t) * 8));
    MR_HIP(hipMalloc(&c->stats, sizeof(unsigned long long) * 16));
    MR_HIP(hipMemset(c->counters, 0, sizeof(uint32_t) * 8));
    MR_HIP(hipMemset(c->stats, 0, sizeof(unsigned long long) * 16));
    MR_HIP(hipMalloc(&c->slot_a, sizeof(int32_t) * N));
    MR_HIP(hipMalloc(&c->mask_a, sizeof(uint32_t) * N));
    MR_HIP(hipMalloc(&c->slot_c, sizeof(int32_t) * N));
    MR_HIP(hipMalloc(&c->pend, sizeof(float) * 18 * N));
    MR_HIP(hipMalloc(&c->noff, sizeof(float) * 2 * (size_t)c->cfg.neighbor_offset_count));
    MR_HIP(hipMalloc(&c->tile_aux, sizeof(float) * 8 * (size_t)c->cfg.light_tile_count * c->cfg.light_tile_size));
    k_neighbor_offsets<<<1, 64, 0, 0>>>(c->cfg.neighbor_offset_count, c->noff);
    MR_HIP(hipDeviceSynchronize());
    *out = c;
    return MIRRES_OK;
}

void mirres_ctx_destroy(mirres_ctx_t* c) {
    if (!c) return;
    void* ptrs[] = {c->any_rays, c->any_hit, c->cl_rays, c->cl_hit, c->counters, c->stats, c->slot_a, c->mask_a, c->slot_c, c->pend, c->noff, c->pool, c->tile_aux, c->ptb};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : c->ev_any) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_cl) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_sync) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_join_pt) (void)hipEventDestroy(c->ev_join_pt);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->pt_stream) (void)hipStreamDestroy(c->pt_stream);
    if (c->ev_join_fin) (void)hipEventDestroy(c->ev_join_fin);
    if (c->ev_join_pt2) (void)hipEventDestroy(c->ev_join_pt2);
    if (c->pt_stream2) (void)hipStreamDestroy(c->pt_stream2);
    for (hipEvent_t e : c->ev_pt) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_band) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_halo_t) (void)hipEventDestroy(e);
    for (int t = 0; t < 2; t++) { if (c->chain_streams[t]) (void)hipStreamDestroy(c->chain_streams[t]); if (c->chain_mem[t]) (void)hipFree(c->chain_mem[t]); }
    if (c->fin_stream) (void)hipStreamDestroy(c->fin_stream);
    for (hipEvent_t e : c->ev_halo) if (e) (void)hipEventDestroy(e);      // created on first use by mirres_render's strip_overlap path (render.hip)
    if (c->halo_stream) (void)hipStreamDestroy(c->halo_stream);
    delete c;
}

int mirres_neighbor_offsets(mirres_ctx_t* ctx, float* out, void* stream) {
    if (!ctx || !out) { set_error("mirres_neighbor_offsets: null"); return MIRRES_E_ARG; }
    MR_HIP(hipMemcpyAsync(out, ctx->noff, sizeof(float) * 2 * (size_t)ctx->cfg.neighbor_offset_count, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MIRRES_OK;
}

int mirres_ctx_stats(mirres_ctx_t* ctx, uint64_t* h_out, int reset) {
    if (!ctx || !h_out) { set_error("mirres_ctx_stats: null"); return MIRRES_E_ARG; }
    MR_HIP(hipDeviceSynchronize());
    MR_HIP(hipMemcpy(h_out, ctx->stats, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
    if (reset) MR_HIP(hipMemset(ctx->stats, 0, sizeof(unsigned long long) * 16));
    return MIRRES_OK;
}
int mirres_ctx_set_instrument(mirres_ctx_t* ctx, int on) { if (!ctx) return MIRRES_E_ARG; ctx->instrument = on; return MIRRES_OK; }
int mirres_ctx_trace_time(mirres_ctx_t* ctx, double* h_ms_any, int* h_n_any, double* h_ms_closest, int* h_n_closest) {
    if (!ctx) return MIRRES_E_ARG;
    MR_HIP(hipDeviceSynchronize());
    double a = 0, c = 0;
    for (size_t i = 0; i + 1 < ctx->ev_any_used; i += 2) { float ms = 0; MR_HIP(hipEventElapsedTime(&ms, ctx->ev_any[i], ctx->ev_any[i + 1])); a += ms; }
    for (size_t i = 0; i + 1 < ctx->ev_cl_used; i += 2) { float ms = 0; MR_HIP(hipEventElapsedTime(&ms, ctx->ev_cl[i], ctx->ev_cl[i + 1])); c += ms; }
    if (h_ms_any) *h_ms_any = a; if (h_n_any) *h_n_any = (int)(ctx->ev_any_used / 2);
    if (h_ms_closest) *h_ms_closest = c; if (h_n_closest) *h_n_closest = (int)(ctx->ev_cl_used / 2);
    ctx->ev_any_used = 0; ctx->ev_cl_used = 0;
    return MIRRES_OK;
}

int mirres_env_make_sampleable(const float* env_tex, int Wc, int Hc, float* pdf, float* cdf, float* mpdf, float* mcdf, void* stream) {
    if (!env_tex || !pdf || !cdf || !mpdf || !mcdf || Wc <= 0 || Hc <= 0) { set_error("mirres_env_make_sampleable: bad argument"); return MIRRES_E_ARG; }
    hipStream_t s = (hipStream_t)stream;
    k_env_weight<<<grid_for((size_t)Wc * Hc, MR_BLOCK), MR_BLOCK, 0, s>>>(env_tex, Wc, Hc, pdf);
    k_env_rows<<<Hc, 64, 0, s>>>(Wc, Hc, pdf, cdf, mpdf);
    k_env_marginal<<<1, 64, 0, s>>>(Hc, mpdf, mcdf);
    MR_LAUNCH_CHECK("make_sampleable");
    return MIRRES_OK;
}

int mirres_light_tiles(mirres_ctx_t* ctx, const float* env_tex, int Wc, int Hc, const float* pdf, const float* cdf, const float* mpdf,
                       const float* mcdf, uint32_t frameIndex, float* light_data, int32_t* light_uv, float* light_inv_pdf, void* stream) {
    if (!ctx || !light_data || !light_inv_pdf) { set_error("mirres_light_tiles: null"); return MIRRES_E_ARG; }
    EnvD E; E.tex = env_tex; E.W = Wc; E.H = Hc; E.pdf = pdf; E.cdf = cdf; E.mpdf = mpdf; E.mcdf = mcdf;
    int total = ctx->cfg.light_tile_count * ctx->cfg.light_tile_size;
    k_light_tiles<<<grid_for(total, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(E, frameIndex, total, total, light_data, light_uv, light_inv_pdf);
    MR_LAUNCH_CHECK("light_tiles");
    return MIRRES_OK;
}

int mirres_restir_initial(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_res_t* res,
                          const float* light_data, const float* light_inv_pdf, uint32_t frameIndex, void* stream) {
    if (!ctx || !bvh || !env || !g || !res) { set_error("mirres_restir_initial: null"); return MIRRES_E_ARG; }
    hipStream_t s = (hipStream_t)stream; const int N = (int)ctx->N; const int grd = grid_for(N, MR_BLOCK);
    MR_HIP(hipMemsetAsync(&ctx->counters[0], 0, sizeof(uint32_t), s));
    const int TS = ctx->cfg.light_tile_count * ctx->cfg.light_tile_size;
    k_tile_aux<false><<<grid_for(TS, MR_BLOCK), MR_BLOCK, 0, s>>>(envh(env), TS, light_data, light_inv_pdf, reinterpret_cast<float4*>(ctx->tile_aux));
    k_initial_gen<false><<<grid_for(N, MR_IGEN_BLOCK), MR_IGEN_BLOCK, 0, s>>>(ctx->cfg, envh(env), gbufd(g), resd(res), light_data, light_inv_pdf, reinterpret_cast<const float4*>(ctx->tile_aux), frameIndex, ctx->fx, N, N, TS, 0, ctx->any_rays,
                                            &ctx->counters[0], ctx->slot_a);
    int rc = trace_any(ctx, bvh, (size_t)N, s); if (rc) return rc;
    k_initial_resolve<<<grd, MR_BLOCK, 0, s>>>(resd(res), N, ctx->slot_a, ctx->any_hit);
    MR_LAUNCH_CHECK("restir_initial");
    return MIRRES_OK;
}

int mirres_restir_temporal(mirres_ctx_t* ctx, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_gbuf_t* prev_g,
                           const mirres_res_t* res, const mirres_res_t* prev_res, const float* motion, uint32_t frameIndex, void* stream) {
    if (!ctx || !env || !g || !prev_g || !res || !prev_res) { set_error("mirres_restir_temporal: null"); return MIRRES_E_ARG; }
    const int N = (int)ctx->N;
    k_temporal<<<grid_for(N, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(ctx->cfg, envh(env), gbufd(g), gbufd(prev_g), resd(res), resd(prev_res), motion,
                                                                            frameIndex, ctx->fx, ctx->fy, N, ctx->y_off);
    MR_LAUNCH_CHECK("restir_temporal");
    return MIRRES_OK;
}

int mirres_restir_spatial(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_res_t* res,
                          const mirres_res_t* prev_res, const float* neighbor_offsets, uint32_t frameIndex, void* stream) {
    return mr::launch_spatial(ctx, bvh, env, g, res, prev_res, neighbor_offsets, frameIndex, (hipStream_t)stream, nullptr, 0u);
}

int mirres_restir_final_vis(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const float* pos, const mirres_res_t* res, float* vis_map, void* stream) {
    if (!ctx || !bvh || !pos || !res || !vis_map) { set_error("mirres_restir_final_vis: null"); return MIRRES_E_ARG; }
    hipStream_t s = (hipStream_t)stream; const int N = (int)ctx->N; const int grd = grid_for(N, MR_BLOCK);
    MR_HIP(hipMemsetAsync(&ctx->counters[0], 0, sizeof(uint32_t), s));
    k_vis_gen<<<grid_for(N, MR_GEN_BLOCK), MR_GEN_BLOCK, 0, s>>>(ctx->cfg.vis_near, pos, resd(res), N, N, ctx->any_rays, &ctx->counters[0], ctx->slot_a);
    int rc = trace_any(ctx, bvh, (size_t)N, s); if (rc) return rc;
    k_vis_resolve<<<grd, MR_BLOCK, 0, s>>>(N, ctx->slot_a, ctx->any_hit, vis_map);
    MR_LAUNCH_CHECK("restir_final_vis");
    return MIRRES_OK;
}

int mirres_restir_eval_final(mirres_ctx_t* ctx, const mirres_env_t* env, const mirres_res_t* res, const float* vis_map, float* final_dir,
                             float* final_dist, float* final_Li, void* stream) {
    if (!ctx || !env || !res || !vis_map || !final_dir || !final_dist || !final_Li) { set_error("mirres_restir_eval_final: null"); return MIRRES_E_ARG; }
    const int N = (int)ctx->N;
    k_eval_final<<<grid_for(N, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(envh(env), resd(res), vis_map, N, final_dir, final_dist, final_Li);
    MR_LAUNCH_CHECK("restir_eval_final");
    return MIRRES_OK;
}

int mirres_restir_eval_final_bwd(mirres_ctx_t* ctx, const mirres_env_t* env, const mirres_res_t* res, const float* vis_map,
                                 const float* grad_final_Li, float* grad_env, void* stream) {
    if (!ctx || !env || !res || !vis_map || !grad_final_Li || !grad_env) { set_error("mirres_restir_eval_final_bwd: null"); return MIRRES_E_ARG; }
    const int N = (int)ctx->N;
    k_eval_final_bwd<<<grid_for(N, MR_BLOCK), MR_BLOCK, 0, (hipStream_t)stream>>>(envh(env), resd(res), vis_map, N, grad_final_Li, grad_env);
    MR_LAUNCH_CHECK("restir_eval_final_bwd");
    return MIRRES_OK;
}

}  // extern "C"
