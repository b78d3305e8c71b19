// ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (see orc_math.hpp header).
// extern "C" surface of the CPU restatement; bound from Python by oracle/oracle.py (ctypes).
// Orchestration restates restir_di_with_pt / run_restir_di_with_pt (nerf/renderer_restir.py:230-550).
#include "orc_kernels.hpp"
#include "orc_matnet.hpp"
#include "orc_dump.hpp"
#include <cstdio>
#include <cstdlib>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

extern "C" {

struct OrcFrame {
    int fx, fy;
    const float *occ, *pos, *normal_depth, *brdf, *ray_dir;
    const int32_t* info; const float* aabb; const float* vert; const int32_t* tri;
    const float* env_tex; int env_w, env_h;
    const float *pdf, *cdf, *mpdf, *mcdf;
    int max_bounce;
    int neighbor_count, initial_light_samples, max_history;   // 0 = the reference's constant (orc_kernels.hpp Config); tests of the runtime configuration set them
};
struct OrcRes { float* light_data; float* light_pdf; int32_t* M; float* weight; };

static inline GBuf gbuf(const OrcFrame* f) { GBuf g; g.fx = f->fx; g.fy = f->fy; g.occ = f->occ; g.pos = f->pos; g.normal_depth = f->normal_depth; g.brdf = f->brdf; g.ray_dir = f->ray_dir; return g; }
static inline Bvh bvh(const OrcFrame* f) { Bvh b; b.info = f->info; b.aabb = f->aabb; b.vert = f->vert; b.tri = f->tri; return b; }
static inline Env env(const OrcFrame* f) { Env e; e.tex = f->env_tex; e.W = f->env_w; e.H = f->env_h; e.pdf = f->pdf; e.cdf = f->cdf; e.mpdf = f->mpdf; e.mcdf = f->mcdf; return e; }
static inline Reservoirs res(const OrcRes* r) { Reservoirs o; o.light_data = r->light_data; o.light_pdf = r->light_pdf; o.M = r->M; o.weight = r->weight; return o; }
static inline Config cfg(const OrcFrame* f) {
    Config c;
    if (f->max_bounce > 0) c.max_bounce = f->max_bounce;
    if (f->neighbor_count > 0) c.neighbor_count = f->neighbor_count;
    if (f->initial_light_samples > 0) c.initial_light_samples = f->initial_light_samples;
    if (f->max_history > 0) c.max_history = f->max_history;
    return c;
}

#define ORC_PIXEL_LOOP(FX, FY, COUNTERS, BODY)                                            \
    {                                                                                     \
        unsigned long long _p = 0, _e = 0, _l = 0, _o = 0;                                \
        _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : _p, _e, _l, _o)")    \
        for (int y = 0; y < (FY); y++) {                                                  \
            TraceCounters tcl = {0, 0, 0, 0}; TraceCounters* tc = &tcl;                   \
            for (int x = 0; x < (FX); x++) { BODY; }                                      \
            _p += tcl.popped; _e += tcl.entered; _l += tcl.leaves; _o += tcl.overflow;    \
        }                                                                                 \
        if (COUNTERS) { (COUNTERS)[0] += _p; (COUNTERS)[1] += _e; (COUNTERS)[2] += _l; (COUNTERS)[3] += _o; } \
    }

// ---- known-answer helpers
uint32_t orc_seed(uint32_t px, uint32_t py, uint32_t n) { return seed_generator(px, py, n); }
float orc_next1d(uint32_t* s) { return next1d(*s); }
uint32_t orc_expand_bits(uint32_t v) { return expand_bits(v); }
uint32_t orc_morton3d(float x, float y, float z) { return morton3d(x, y, z); }
void orc_oct_encode(const float* n, float* out) { f2 r = oct_encode(mk3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; }
void orc_oct_decode(const float* f, float* out) { f3 r = oct_decode(mk2(f[0], f[1])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

// ---- BVH
int orc_bvh_build(const float* vert, int V, const int32_t* tri, int T, int32_t* info, float* aabb, int32_t* sorted, int* max_height) {
    if (T < 2) return -1;
    BuildStats st; bvh_build(vert, V, tri, T, info, aabb, sorted, &st);
    if (max_height) *max_height = st.max_height;
    return 0;
}
// the seven build kernels one by one (driven by the reference's own update_bvh in tests/golden/gen_reference_loop.py)
void orc_bvh_elements(const float* vert, const int32_t* tri, int T, int32_t* ele_prim, float* ele) { bvh_elements(vert, tri, T, ele_prim, ele); }
void orc_bvh_morton(int T, const float* gmin, const float* gmax, const float* ele, int32_t* codes) { bvh_morton(T, gmin, gmax, ele, codes); }
void orc_bvh_radix_sort(int T, int32_t* a, int32_t* b) { bvh_radix_sort(T, a, b); }
void orc_bvh_hierarchy(int T, const int32_t* ele_prim, const float* ele, const int32_t* sorted, int32_t* info, float* aabb, int32_t* cons) { bvh_hierarchy(T, ele_prim, ele, sorted, info, aabb, cons); }
void orc_bvh_heights(int T, const int32_t* cons, int32_t* heights) { bvh_heights(T, cons, heights); }
void orc_bvh_bbox_pass(int T, int eh, const int32_t* info, float* aabb, const int32_t* cons) { bvh_bbox_pass(T, eh, info, aabb, cons); }
void orc_bvh_set_root(const int32_t* info, float* aabb) { bvh_set_root(info, aabb); }
// rays [n,8] = (ox,oy,oz,tmin, dx,dy,dz,tmax). counters [n,4] optional (popped, entered, leaves, overflow).
void orc_trace(const int32_t* info, const float* aabb, const float* vert, const int32_t* tri, const float* rays, int n, int want_normal,
               int32_t* hit, float* t, float* pos, float* normal, int32_t* prim, uint32_t* counters) {
    Bvh B = {info, aabb, vert, tri};
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) {
        const float* r = rays + 8 * (size_t)i;
        TraceCounters tc = {0, 0, 0, 0};
        HitResult h = bvh_hit(B, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7], want_normal != 0, &tc);
        hit[i] = h.hit ? 1 : 0; t[i] = h.t; prim[i] = h.prim;
        if (pos) st3(pos, i, h.pos);
        if (normal) st3(normal, i, h.normal);
        if (counters) { counters[4 * (size_t)i] = tc.popped; counters[4 * (size_t)i + 1] = tc.entered; counters[4 * (size_t)i + 2] = tc.leaves; counters[4 * (size_t)i + 3] = tc.overflow; }
    }
}

// deepest the reference's 64-entry traversal stack gets for each ray (bvh_hit, helperDi.slang:197-274), and the depth of the LBVH itself
void orc_trace_stack_depth(const int32_t* info, const float* aabb, const float* vert, const int32_t* tri, const float* rays, int n, int want_normal, uint32_t* max_count) {
    Bvh B = {info, aabb, vert, tri};
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) {
        const float* r = rays + 8 * (size_t)i;
        TraceCounters tc = {0, 0, 0, 0, 1};
        (void)bvh_hit(B, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7], want_normal != 0, &tc);
        max_count[i] = tc.max_count;
    }
}

// ---- environment
void orc_make_sampleable(const float* tex, int W, int H, float* pdf, float* cdf, float* mpdf, float* mcdf) { make_sampleable(tex, W, H, pdf, cdf, mpdf, mcdf); }
void orc_neighbor_offsets(int count, float* out) { neighbor_offsets(count, out); }
void orc_neighbor_offsets_raw(int count, float* out) { neighbor_offsets_raw(count, out); }
void orc_env_weights(const float* tex, int W, int H, float* weight) { env_weights(tex, W, H, weight); }
void orc_distribution2d(int W, int H, float* pdf, float* cdf) { distribution2d(W, H, pdf, cdf); }
void orc_env_le(const float* tex, int W, int H, const float* dirs, int n, float* out) {
    for (int i = 0; i < n; i++) st3(out, i, env_le(ld3(dirs, i), tex, W, H));
}
void orc_light_tiles(const OrcFrame* f, uint32_t frameIndex, int tile_count, int tile_size, float* light_data, int32_t* light_uv, float* light_pdf) {
    light_tiles(env(f), frameIndex, tile_count, tile_size, light_data, light_uv, light_pdf);
}

// ---- ReSTIR passes
void orc_initial(const OrcFrame* f, const OrcRes* r, const float* tile_data, const float* tile_pdf, uint32_t frameIndex, unsigned long long* counters) {
    Config C = cfg(f); Bvh B = bvh(f); Env E = env(f); GBuf G = gbuf(f); Reservoirs R = res(r);
    ORC_PIXEL_LOOP(f->fx, f->fy, counters, initial_pixel(C, B, E, G, R, tile_data, tile_pdf, frameIndex, x, y, tc));
}
void orc_temporal(const OrcFrame* f, const OrcRes* r, const OrcRes* prev, const float* p_occ, const float* p_nd, const float* p_brdf, const float* p_rd,
                  const float* motion, uint32_t frameIndex) {
    Config C = cfg(f); Env E = env(f); GBuf G = gbuf(f); Reservoirs R = res(r), PR = res(prev);
    PrevGBuf P = {p_occ, p_nd, p_brdf, p_rd};
    unsigned long long* none = nullptr;
    ORC_PIXEL_LOOP(f->fx, f->fy, none, (void)tc; temporal_pixel(C, E, G, P, R, PR, motion, frameIndex, x, y));
}
void orc_spatial(const OrcFrame* f, const OrcRes* r, const OrcRes* prev, const float* neighborOffsets, uint32_t frameIndex, unsigned long long* counters) {
    Config C = cfg(f); Bvh B = bvh(f); Env E = env(f); GBuf G = gbuf(f); Reservoirs R = res(r), PR = res(prev);
    ORC_PIXEL_LOOP(f->fx, f->fy, counters, spatial_pixel(C, B, E, G, R, PR, neighborOffsets, frameIndex, x, y, tc));
}
void orc_final_vis(const OrcFrame* f, const OrcRes* r, float* vis, unsigned long long* counters) {
    Config C = cfg(f); Bvh B = bvh(f); GBuf G = gbuf(f); Reservoirs R = res(r);
    ORC_PIXEL_LOOP(f->fx, f->fy, counters, final_vis_pixel(C, B, G, R, vis, x, y, tc));
}
void orc_eval_final(const OrcFrame* f, const OrcRes* r, const float* vis, float* fdir, float* fdist, float* fLi) {
    Env E = env(f); Reservoirs R = res(r);
    size_t N = (size_t)f->fx * f->fy;
#pragma omp parallel for
    for (long long i = 0; i < (long long)N; i++) eval_final_pixel(E, R, vis, fdir, fdist, fLi, (size_t)i);
}
void orc_final_shading(const OrcFrame* f, const float* normal, const float* kd, const float* rs, const float* fdir, const float* fdist, const float* fLi,
                       float* color, float* diff_light, float* spec_light) {
    Env E = env(f);
    size_t N = (size_t)f->fx * f->fy;
#pragma omp parallel for
    for (long long i = 0; i < (long long)N; i++) final_shading_pixel(E, f->occ, normal, f->ray_dir, kd, rs, fdir, fdist, fLi, color, diff_light, spec_light, (size_t)i);
}

struct OrcPath {
    const float *occ, *pos, *normal, *ray_dir, *kd, *rs;
    float* prd; float *new_pos, *new_ray_d, *new_occ, *new_normal;
};
static inline PathBufs pathbufs(const OrcPath* p) { PathBufs b; b.occ = p->occ; b.pos = p->pos; b.normal = p->normal; b.ray_dir = p->ray_dir; b.kd = p->kd; b.rs = p->rs; b.prd = p->prd; b.new_pos = p->new_pos; b.new_ray_d = p->new_ray_d; b.new_occ = p->new_occ; b.new_normal = p->new_normal; return b; }

void orc_new_dir(const OrcFrame* f, const OrcPath* p, uint32_t frameIndex, uint32_t bounce_count, unsigned long long* counters) {
    Config C = cfg(f); Bvh B = bvh(f); PathBufs P = pathbufs(p);
    ORC_PIXEL_LOOP(f->fx, f->fy, counters, new_dir_pixel(C, B, P, f->fx, frameIndex, bounce_count, x, y, tc));
}
void orc_bounce(const OrcFrame* f, const OrcPath* p, uint32_t frameIndex, uint32_t bounce_count, float* color, float* diff_color, float* spec_color,
                unsigned long long* counters) {
    Config C = cfg(f); Bvh B = bvh(f); Env E = env(f); PathBufs P = pathbufs(p);
    ORC_PIXEL_LOOP(f->fx, f->fy, counters, bounce_pixel(C, B, E, P, f->fx, frameIndex, bounce_count, color, diff_color, spec_color, x, y, tc));
}
// process_normal_ao (EAWDenoise.slang:591-651): offsets -4 .. 3, x-offset outer / y-offset inner, foreground neighbours only
void orc_normal_ao(int fx, int fy, const float* occ, const float* normal, float* out) {
#pragma omp parallel for
    for (int y = 0; y < fy; y++)
        for (int x = 0; x < fx; x++) {
            const int pi = y * fx + x;
            float v = 0.f;
            if (!(occ[pi] < 0.1f)) {
                const float nx = normal[3 * pi], ny = normal[3 * pi + 1], nz = normal[3 * pi + 2];
                float sum = 0.f; int count = 0;
                for (int i = -4; i < 4; i++)
                    for (int j = -4; j < 4; j++) {
                        const int ux = x + i, uy = y + j;
                        if (ux < 0 || uy < 0 || ux >= fx || uy >= fy) continue;
                        const int q = uy * fx + ux;
                        if (occ[q] < 0.1f) continue;
                        float d = normal[3 * q] * nx + normal[3 * q + 1] * ny + normal[3 * q + 2] * nz;
                        d = d > 0.0f ? d : 0.0f; d = d < 1.0f ? d : 1.0f;
                        sum += d; count++;
                    }
                float w = (1.f - sum / (float)count) * 50.f;
                v = w < 0.f ? 0.f : (w > 1.f ? 1.f : w);
            }
            out[3 * pi] = out[3 * pi + 1] = out[3 * pi + 2] = v;
        }
}

void orc_eaw(int fx, int fy, int stepWidth, float c_phi, float n_phi, float p_phi, const float* occ, const float* color, const float* normal,
             const float* pos, float* out) {
#pragma omp parallel for
    for (int y = 0; y < fy; y++)
        for (int x = 0; x < fx; x++) eaw_pixel(fx, fy, stepWidth, c_phi, n_phi, p_phi, occ, color, normal, pos, out, x, y);
}

// ---- bilateral denoiser (nerf/renderutils/ops.py:164-211 + c_src/denoising.cu:14-130), the --use_bi_de branch of run_restir_di_with_pt
// (renderer_restir.py:529-541). mode 0: out4 = (sum w col, max(sum w, 1e-4)); mode 1: col_grad3 = sum w' grad4.xyz (transposed depth term).
void orc_bilateral(int fx, int fy, float sigma, const float* col, const float* nrm_in, const float* zdz, const float* grad4, int mode, float* out) {
    const size_t n = (size_t)fx * fy;
    std::vector<float> nrm(3 * n);
    for (size_t i = 0; i < n; i++) {                        // safe_normalize (ops.py:164-169)
        float x = nrm_in[3 * i], y = nrm_in[3 * i + 1], z = nrm_in[3 * i + 2];
        float len = std::sqrt(std::max((x * x + y * y) + z * z, 1e-20f));
        nrm[3 * i] = x / len; nrm[3 * i + 1] = y / len; nrm[3 * i + 2] = z / len;
    }
    const float variance = sigma * sigma;
    const int rad = 2 * (int)std::ceil(sigma * 2.5f) + 1;
#pragma omp parallel for
    for (int y = 0; y < fy; y++)
        for (int x = 0; x < fx; x++) {
            const size_t pi = (size_t)y * fx + x;
            const float cz = zdz[2 * pi], cdz = zdz[2 * pi + 1];
            float aw = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
            for (int dy = -rad; dy <= rad; ++dy)
                for (int dx = -rad; dx <= rad; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || xx < 0 || yy >= fy || xx >= fx) continue;
                    const size_t qi = (size_t)yy * fx + xx;
                    const float dist_sqr = (float)(dx * dx + dy * dy), dist = std::sqrt(dist_sqr);
                    const float w_xy = mrf_exp(-dist_sqr / (2.0f * variance));
                    const float nd = (nrm[3 * qi] * nrm[3 * pi] + nrm[3 * qi + 1] * nrm[3 * pi + 1]) + nrm[3 * qi + 2] * nrm[3 * pi + 2];
                    const float w_normal = mrf_pow2k(std::min(std::max(nd, 0.0001f), 1.0f), 7);
                    const float den = std::max((mode == 1 ? zdz[2 * qi + 1] : cdz) * dist, 0.0001f);
                    const float w_depth = mrf_exp(-(std::fabs(zdz[2 * qi] - cz) / den));
                    const float w = w_xy * w_normal * w_depth;
                    if (mode == 1) { ax += grad4[4 * qi] * w; ay += grad4[4 * qi + 1] * w; az += grad4[4 * qi + 2] * w; }
                    else { ax += col[3 * qi] * w; ay += col[3 * qi + 1] * w; az += col[3 * qi + 2] * w; aw += w; }
                }
            if (mode == 1) { out[3 * pi] = ax; out[3 * pi + 1] = ay; out[3 * pi + 2] = az; }
            else { out[4 * pi] = ax; out[4 * pi + 1] = ay; out[4 * pi + 2] = az; out[4 * pi + 3] = std::max(aw, 0.0001f); }
        }
}

// ---- material field
struct OrcMatNet {
    const uint16_t* params_f16;  // [total_entries*2] fp16 bits
    const float *w0, *w1, *w2;
    float aabb_min[3], aabb_max[3], mn[6], mx[6];
};
int orc_hashgrid_layout(uint32_t* offsets /*[17]*/, uint32_t* resolutions /*[16]*/, float* scales /*[16]*/) {
    HashGridCfg c; uint32_t total = 0;
    std::vector<GridLevel> L = grid_levels(c, &total);
    for (int i = 0; i < 16; i++) { offsets[i] = L[i].offset; resolutions[i] = L[i].resolution; scales[i] = L[i].scale; }
    offsets[16] = total;
    return (int)total;
}
void orc_f32_to_f16_array(const float* in, uint16_t* out, long long n) {
#pragma omp parallel for
    for (long long i = 0; i < n; i++) out[i] = f32_to_f16(in[i]);
}
static inline MatNet matnet(const OrcMatNet* m) {
    MatNet M; HashGridCfg c; M.levels = grid_levels(c, nullptr); M.params = m->params_f16; M.w0 = m->w0; M.w1 = m->w1; M.w2 = m->w2;
    for (int i = 0; i < 3; i++) { M.aabb_min[i] = m->aabb_min[i]; M.aabb_max[i] = m->aabb_max[i]; }
    for (int i = 0; i < 6; i++) { M.mn[i] = m->mn[i]; M.mx[i] = m->mx[i]; }
    return M;
}
void orc_hashgrid_encode(const OrcMatNet* m, const float* x01, int n, uint16_t* out /*[n,32] fp16 bits*/) {
    MatNet M = matnet(m);
#pragma omp parallel for
    for (int i = 0; i < n; i++) hashgrid_encode(M.levels, M.params, x01 + 3 * (size_t)i, out + 32 * (size_t)i);
}
void orc_matnet(const OrcMatNet* m, const float* pos, int n, float* out /*[n,6]*/) {
    MatNet M = matnet(m);
#pragma omp parallel for
    for (int i = 0; i < n; i++) matnet_eval(M, pos + 3 * (size_t)i, out + 6 * (size_t)i);
}

// ---- whole frame: run_restir_di_with_pt (renderer_restir.py:473-550) + restir_di_with_pt (:230-471)
// env_map [Hc,Wc,3] un-flipped (as the caller passes it); occ is modified in place like the reference (:484-485).
// outs: 6 x [N,3] (final_color, den_diffuse, den_spec, den_indirect, den_indirect_diff, den_indirect_spec).
struct OrcRenderArgs {
    int fx, fy, spp; uint32_t random_offset; int max_bounce;
    int use_scale; float scale[3];
    const int32_t* info; const float* aabb; const float* vert; const int32_t* tri;
    const float* env_map; int env_w, env_h;
    float* occ; const float *normal, *depth, *kd, *rs, *ray_dir, *pos;
    const OrcMatNet* mat;  // may be null -> constant material (const_kd, const_rs) at indirect hits
    float const_kd[3]; float const_rs[2];
    int denoise_iter, step_width; float c_phi, n_phi, p_phi;
    float* outs[6];
    unsigned long long* counters;  // [4] optional traversal totals
    unsigned long long* ray_count; // [1] optional total rays traced
    float* avg_direct;  // optional [N,3] un-denoised mean colour (total_color / spp)
};

// second half of run_restir_di_with_pt (renderer_restir.py:507-549) on the six raw sums (total colour, diffuse, specular, indirect colour,
// indirect diffuse, indirect specular; divided by spp IN PLACE): averages, the five a-trous runs, composite, background = 1, nan_to_num.
// Pinned by tests/golden/ref_python.npz (the reference's own function over prepared sums).
void orc_eaw(int fx, int fy, int stepWidth, float c_phi, float n_phi, float p_phi, const float* occ, const float* color, const float* normal,
             const float* pos, float* out);
void orc_finish(int fx, int fy, int spp, const float* occ, const float* normal, const float* pos, const float* kd, const float* rs, int denoise_iter, int step_width,
                float c_phi, float n_phi, float p_phi, float* const* sums, float* const* outs) {
    const size_t N = (size_t)fx * fy;
    const float inv = (float)spp;
    std::vector<float> comb(3 * N);
    for (size_t i = 0; i < 3 * N; i++) {
        for (int k = 0; k < 6; k++) sums[k][i] /= inv;
        comb[i] = sums[4][i] + sums[5][i];
    }
    auto denoise = [&](const float* in, float* out) {
        std::vector<float> cur(in, in + 3 * N), nxt(3 * N);
        float swf = (float)step_width;
        for (int it = 0; it < denoise_iter; it++) {
            const int sw = (int)swf;  // Denoising.py: stepWidth /= 2 (float), int(stepWidth) at launch
            orc_eaw(fx, fy, sw, c_phi, n_phi, p_phi, occ, cur.data(), normal, pos, nxt.data());
            cur.swap(nxt); swf = swf / 2;
        }
        std::memcpy(out, cur.data(), sizeof(float) * 3 * N);
    };
    denoise(sums[1], outs[1]); denoise(sums[2], outs[2]); denoise(comb.data(), outs[3]); denoise(sums[4], outs[4]); denoise(sums[5], outs[5]);
    for (size_t i = 0; i < N; i++) {
        float m = rs[2 * i + 1];
        for (int k = 0; k < 3; k++) {
            float d = kd[3 * i + k] * (1.0f - m);
            float v = d * outs[1][3 * i + k] + outs[2][3 * i + k] + outs[3][3 * i + k];
            if (occ[i] <= 0.1f) v = 1.0f;
            if (std::isnan(v)) v = 0.f;                      // torch.nan_to_num(x, 0.0): nan->0, +-inf -> +-FLT_MAX
            else if (std::isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
            outs[0][3 * i + k] = v;
        }
    }
}

static void accumulate(std::vector<float>& a, const std::vector<float>& b) {
#pragma omp parallel for
    for (long long i = 0; i < (long long)a.size(); i++) a[i] += b[i];
}

// ReSTIR constants of the NEXT orc_render calls (0 = the reference's value): the reference fixes them at compile time, the product takes them as runtime
// configuration (mirres_config_t) — frame-level tests of that configuration set the same numbers here
static int g_render_neighbor_count = 0, g_render_initial_light_samples = 0, g_render_max_history = 0;
void orc_set_render_constants(int neighbor_count, int initial_light_samples, int max_history) {
    g_render_neighbor_count = neighbor_count; g_render_initial_light_samples = initial_light_samples; g_render_max_history = max_history;
}
long long orc_set_dead_ray_override(int mode) { g_dead_ray_override = mode; const long long n = g_dead_ray_count; g_dead_ray_count = 0; return n; }   // orc_kernels.hpp: test hook, -1 = off; returns the rays answered by the hook since the last call
int orc_render(const OrcRenderArgs* A) {
    const int fx = A->fx, fy = A->fy; const size_t N = (size_t)fx * fy;
    Config C; if (A->max_bounce > 0) C.max_bounce = A->max_bounce;
    if (g_render_neighbor_count > 0) C.neighbor_count = g_render_neighbor_count;
    if (g_render_initial_light_samples > 0) C.initial_light_samples = g_render_initial_light_samples;
    if (g_render_max_history > 0) C.max_history = g_render_max_history;
    // run_restir_di_with_pt :484-486
    for (size_t i = 0; i < N; i++) if (A->occ[i] <= 0.5f) A->occ[i] = 0.f;
    std::vector<float> ray_dir(3 * N);
    for (size_t i = 0; i < N; i++) {  // F.normalize(eps=1e-6)
        f3 d = ld3(A->ray_dir, i); float l = fmaxf(sqrtf(dot(d, d)), 1e-6f); st3(ray_dir.data(), i, mk3(d.x / l, d.y / l, d.z / l));
    }
    // restir_di_with_pt :279-287
    std::vector<float> nd(4 * N), brdf(3 * N);
    for (size_t i = 0; i < N; i++) {
        nd[4 * i] = A->normal[3 * i]; nd[4 * i + 1] = A->normal[3 * i + 1]; nd[4 * i + 2] = A->normal[3 * i + 2]; nd[4 * i + 3] = A->depth[i];
        const float* k = A->kd + 3 * i; float m = A->rs[2 * i + 1], r = A->rs[2 * i];
        brdf[3 * i] = (k[0] * 0.2126f + k[1] * 0.7152f) + k[2] * 0.0722f;
        brdf[3 * i + 1] = (m * 0.2126f + m * 0.7152f) + m * 0.0722f;
        float a = fminf(fmaxf(r, 0.01f), 1.f);
        brdf[3 * i + 2] = a * a;
    }
    // env flip :305-311 + make_sampleable
    const int W = A->env_w, H = A->env_h;
    std::vector<float> tex(3 * (size_t)W * H), pdf((size_t)W * H), cdf((size_t)(W + 1) * H), mpdf(H), mcdf(H + 1);
    for (int y = 0; y < H; y++) std::memcpy(&tex[3 * (size_t)y * W], A->env_map + 3 * (size_t)(H - 1 - y) * W, sizeof(float) * 3 * W);
    make_sampleable(tex.data(), W, H, pdf.data(), cdf.data(), mpdf.data(), mcdf.data());
    Env E = {tex.data(), W, H, pdf.data(), cdf.data(), mpdf.data(), mcdf.data()};
    Bvh B = {A->info, A->aabb, A->vert, A->tri};
    GBuf G; G.fx = fx; G.fy = fy; G.occ = A->occ; G.pos = A->pos; G.normal_depth = nd.data(); G.brdf = brdf.data(); G.ray_dir = ray_dir.data();
    std::vector<float> noff(2 * (size_t)C.neighbor_offset_count); neighbor_offsets(C.neighbor_offset_count, noff.data());
    const size_t TS = (size_t)C.light_tile_count * C.light_tile_size;
    std::vector<float> tile_data(3 * TS), tile_pdf(TS); std::vector<int32_t> tile_uv(2 * TS);
    // reservoir ping-pong
    std::vector<float> r_ld[2], r_pdf[2], r_w[2]; std::vector<int32_t> r_M[2];
    for (int k = 0; k < 2; k++) { r_ld[k].assign(3 * N, 0.f); r_pdf[k].assign(N, 0.f); r_w[k].assign(N, 0.f); r_M[k].assign(N, 0); }
    Reservoirs RA = {r_ld[0].data(), r_pdf[0].data(), r_M[0].data(), r_w[0].data()};
    Reservoirs RB = {r_ld[1].data(), r_pdf[1].data(), r_M[1].data(), r_w[1].data()};
    Reservoirs *reservoirs = &RA, *prev_reservoirs = &RB;
    std::vector<float> vis(N, 1.f), fdir(3 * N), fdist(N), fLi(3 * N), color(3 * N), cdiff(3 * N), cspec(3 * N);
    std::vector<float> total_color(3 * N, 0.f), total_diff(3 * N, 0.f), total_spec(3 * N, 0.f);
    std::vector<float> total_color_1(3 * N, 0.f), total_diff_1(3 * N, 0.f), total_spec_1(3 * N, 0.f), color_1(3 * N, 0.f), cdiff_1(3 * N, 0.f), cspec_1(3 * N, 0.f);
    std::vector<float> prd(5 * N, 0.f), new_pos(3 * N, 0.f), new_ray_d(3 * N, 0.f), new_occ(N, 0.f), new_normal(3 * N, 0.f), new_kd(3 * N, 0.f), new_rs(2 * N, 0.f);
    std::vector<float> tmp_pos(3 * N, 0.f), tmp_ray_d(3 * N, 0.f), tmp_occ(N, 0.f), tmp_normal(3 * N, 0.f);
    MatNet M; if (A->mat) M = matnet(A->mat);
    unsigned long long cnt[4] = {0, 0, 0, 0};
    const uint32_t passes = 20;  // mTotalRISPasses
    bool have_prev = false;

    auto run_matnet = [&](const std::vector<float>& occm, const std::vector<float>& posm) {  // :398-408 / :428-438
#pragma omp parallel for
        for (long long i = 0; i < (long long)N; i++) {
            if (occm[i] >= 0.5f) {
                float o[6] = {A->const_kd[0], A->const_kd[1], A->const_kd[2], 0.f, A->const_rs[0], A->const_rs[1]};
                if (A->mat) matnet_eval(M, &posm[3 * i], o);
                new_kd[3 * i] = o[0]; new_kd[3 * i + 1] = o[1]; new_kd[3 * i + 2] = o[2];
                new_rs[2 * i] = o[4]; new_rs[2 * i + 1] = o[5];
                if (A->use_scale) for (int k = 0; k < 3; k++) new_kd[3 * i + k] = new_kd[3 * i + k] * A->scale[k];
            }
            if (A->use_scale) for (int k = 0; k < 3; k++) new_kd[3 * i + k] = fminf(fmaxf(new_kd[3 * i + k], 0.f), 1.f);
        }
    };

    for (int i = 0; i < A->spp; i++) {
        uint32_t pass = 0;
        uint32_t frameIndex = A->random_offset + passes * (uint32_t)i + pass;
        light_tiles(E, frameIndex, C.light_tile_count, C.light_tile_size, tile_data.data(), tile_uv.data(), tile_pdf.data());
        pass += 2;
        frameIndex = A->random_offset + passes * (uint32_t)i + pass;
        { Reservoirs R = *reservoirs; ORC_PIXEL_LOOP(fx, fy, cnt, initial_pixel(C, B, E, G, R, tile_data.data(), tile_pdf.data(), frameIndex, x, y, tc)); }
        pass += 1;
        frameIndex = A->random_offset + passes * (uint32_t)i + pass;
        if (i > 0) {
            // prev_* G-buffers equal the current ones from spp 1 on (:462-465)
            PrevGBuf P = {G.occ, G.normal_depth, G.brdf, G.ray_dir};
            Reservoirs R = *reservoirs, PR = *prev_reservoirs;
            unsigned long long* none = nullptr;
            ORC_PIXEL_LOOP(fx, fy, none, (void)tc; temporal_pixel(C, E, G, P, R, PR, nullptr, frameIndex, x, y));
            pass += 1;
        }
        (void)have_prev;
        frameIndex = A->random_offset + passes * (uint32_t)i + pass;
        std::swap(reservoirs, prev_reservoirs);
        { Reservoirs R = *reservoirs, PR = *prev_reservoirs; ORC_PIXEL_LOOP(fx, fy, cnt, spatial_pixel(C, B, E, G, R, PR, noff.data(), frameIndex, x, y, tc)); }
        pass += 1;
        { Reservoirs R = *reservoirs; ORC_PIXEL_LOOP(fx, fy, cnt, final_vis_pixel(C, B, G, R, vis.data(), x, y, tc)); }
        {
            Reservoirs R = *reservoirs;
#pragma omp parallel for
            for (long long p = 0; p < (long long)N; p++) {
                eval_final_pixel(E, R, vis.data(), fdir.data(), fdist.data(), fLi.data(), (size_t)p);
                final_shading_pixel(E, A->occ, A->normal, ray_dir.data(), A->kd, A->rs, fdir.data(), fdist.data(), fLi.data(), color.data(), cdiff.data(), cspec.data(), (size_t)p);
            }
        }
        frameIndex = A->random_offset + passes * (uint32_t)i + pass;
        {
            PathBufs P; P.occ = A->occ; P.pos = A->pos; P.normal = A->normal; P.ray_dir = ray_dir.data(); P.kd = A->kd; P.rs = A->rs; P.prd = prd.data();
            P.new_pos = new_pos.data(); P.new_ray_d = new_ray_d.data(); P.new_occ = new_occ.data(); P.new_normal = new_normal.data();
            ORC_PIXEL_LOOP(fx, fy, cnt, new_dir_pixel(C, B, P, fx, frameIndex, 0, x, y, tc));
        }
        pass += 5;
        // indirect vertices: the reference unrolls exactly two (bounce_count 1 and 2); generalised to max_bounce
        std::vector<float>*cur_pos = &new_pos, *cur_rd = &new_ray_d, *cur_occ = &new_occ, *cur_n = &new_normal;
        std::vector<float>*nxt_pos = &tmp_pos, *nxt_rd = &tmp_ray_d, *nxt_occ = &tmp_occ, *nxt_n = &tmp_normal;
        for (int b = 1; b <= C.max_bounce; b++) {
            run_matnet(*cur_occ, *cur_pos);
            frameIndex = A->random_offset + passes * (uint32_t)i + pass;
            PathBufs P; P.occ = cur_occ->data(); P.pos = cur_pos->data(); P.normal = cur_n->data(); P.ray_dir = cur_rd->data(); P.kd = new_kd.data(); P.rs = new_rs.data();
            P.prd = prd.data(); P.new_pos = nxt_pos->data(); P.new_ray_d = nxt_rd->data(); P.new_occ = nxt_occ->data(); P.new_normal = nxt_n->data();
            ORC_PIXEL_LOOP(fx, fy, cnt, bounce_pixel(C, B, E, P, fx, frameIndex, (uint32_t)b, color_1.data(), cdiff_1.data(), cspec_1.data(), x, y, tc));
            accumulate(total_color_1, color_1); accumulate(total_diff_1, cdiff_1); accumulate(total_spec_1, cspec_1);
            pass += 5;
            std::swap(cur_pos, nxt_pos); std::swap(cur_rd, nxt_rd); std::swap(cur_occ, nxt_occ); std::swap(cur_n, nxt_n);
        }
        std::swap(reservoirs, prev_reservoirs);
        accumulate(total_color, color); accumulate(total_diff, cdiff); accumulate(total_spec, cspec);
    }
    if (A->counters) for (int k = 0; k < 4; k++) A->counters[k] += cnt[k];
    float* sums[6] = {total_color.data(), total_diff.data(), total_spec.data(), total_color_1.data(), total_diff_1.data(), total_spec_1.data()};
    orc_finish(fx, fy, A->spp, A->occ, A->normal, A->pos, A->kd, A->rs, A->denoise_iter, A->step_width, A->c_phi, A->n_phi, A->p_phi, sums, A->outs);
    if (A->avg_direct) std::memcpy(A->avg_direct, total_color.data(), sizeof(float) * 3 * N);
    return 0;
}

// ---- probes of the two BRDF libraries (orc_brdf.hpp), one call per array of inputs: what tests/test_brdf_invariants.py holds against the published formulas
// (Walter et al. 2007 GGX, Heitz 2014 Smith masking, Schlick Fresnel, Falcor's lobe selection) written out independently in float64 numpy
void orc_sh_lobes(int n, const float* kd, const float* rough, const float* metal, const float* ray_dir, const float* normal, float* pD, float* pS, float* alpha, float* specular) {
    for (int i = 0; i < n; i++) {
        sh::Lobes L = sh::lobes(ld3(kd, i), rough[i], metal[i], ld3(ray_dir, i), ld3(normal, i));
        pD[i] = L.pD; pS[i] = L.pS; alpha[i] = L.alpha; st3(specular, i, L.specular);
    }
}
// FalcorBRDF_eval / FalcorBRDF_evalPdf (activeLobes = true, allowDeltaEval = false) and the two lobes on their own, local frame (z = normal)
void orc_sh_eval(int n, const float* pD, const float* pS, const float* alpha, const float* spec_albedo, const float* diff_albedo, const float* wo, const float* wi,
                 float* f /*[n,3]*/, float* pdf /*[n]*/, float* spec_f /*[n,3]*/, float* spec_pdf /*[n]*/, float* diff_light /*[n]*/) {
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        f3 o = ld3(wo, i), w = ld3(wi, i);
        st3(f, i, sh::falcor_eval(pD[i], pS[i], alpha[i], ld3(spec_albedo, i), ld3(diff_albedo, i), o, w));
        pdf[i] = sh::falcor_eval_pdf(pD[i], pS[i], o, w, alpha[i]);
        st3(spec_f, i, sh::specular_eval(o, w, ld3(spec_albedo, i), alpha[i]));
        spec_pdf[i] = sh::specular_eval_pdf(o, w, alpha[i]);
        diff_light[i] = sh::diffuse_light(o, w).x;
    }
}
// FalcorBRDF_sample (with_weight) / FalcorBRDF_sample_no_weight from generator state sg[i]; also returns the state afterwards (draw count) and the lobe choice u
void orc_sh_sample(int n, const uint32_t* sg_in, float pD, float pS, float alpha, const float* spec_albedo, const float* diff_albedo, const float* wo, int with_weight,
                   float* wi /*[n,3]*/, float* pdf, uint32_t* specular_bounce, float* weight /*[n,3]*/, int32_t* valid, uint32_t* sg_out, float* u_select) {
    const f3 sa = mk3(spec_albedo[0], spec_albedo[1], spec_albedo[2]), da = mk3(diff_albedo[0], diff_albedo[1], diff_albedo[2]), o = mk3(wo[0], wo[1], wo[2]);
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        uint32_t sg = sg_in[i], probe = sg_in[i];
        u_select[i] = next1d(probe);
        f3 w, wt; float p; uint32_t sb;
        valid[i] = sh::falcor_sample(pD, pS, o, w, p, sb, wt, sg, alpha, sa, da, with_weight != 0) ? 1 : 0;
        st3(wi, i, w); pdf[i] = p; specular_bounce[i] = sb; st3(weight, i, wt); sg_out[i] = sg;
    }
}
// utils/brdf.slang (reservoir target function and candidate pdf): evalBRDF, evalPdfBRDF(specularOnly = false), sampleBRDF, world space
void orc_rt_eval(int n, const float* L, const float* V, const float* N, const float* alpha, const float* wd, const float* ws, float* f, float* pdf) {
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        f[i] = rt::eval_brdf(ld3(L, i), ld3(V, i), ld3(N, i), alpha[i], wd[i], ws[i]);
        pdf[i] = rt::eval_pdf_brdf(ld3(L, i), ld3(V, i), ld3(N, i), alpha[i], wd[i], ws[i]);
    }
}
void orc_rt_sample(int n, const float* xi /*[n,3]*/, const float* V, const float* N, float alpha, float wd, float ws, float* dir /*[n,3]*/, int32_t* valid) {
    const f3 v = mk3(V[0], V[1], V[2]), nn = mk3(N[0], N[1], N[2]);
#pragma omp parallel for
    for (int i = 0; i < n; i++) {
        f3 d; valid[i] = rt::sample_brdf(ld3(xi, i), d, v, nn, alpha, wd, ws) ? 1 : 0; st3(dir, i, d);
    }
}
void orc_sh_frame(const float* n, float* x, float* y) { sh::Frame f = sh::create_frame(mk3(n[0], n[1], n[2])); x[0] = f.x.x; x[1] = f.x.y; x[2] = f.x.z; y[0] = f.y.x; y[1] = f.y.y; y[2] = f.y.z; }

// ---- nerf/render_dump.py (BASELINE configs[0]: direct lighting over a fixed lat-long light set, no ReSTIR)
void orc_occluded_front(const int32_t* info, const float* aabb, const float* vert, const int32_t* tri, const float* rays, int n, int32_t* hit) {
    Bvh B = {info, aabb, vert, tri};
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) { const float* r = rays + 8 * (size_t)i; hit[i] = bvh_occluded_front(B, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7]) ? 1 : 0; }
}
void orc_dump_light_rgbs(const float* env, int H, int W, const float* dirs, int L, float* out) {
    for (int l = 0; l < L; l++) st3(out, l, dump_light_rgb(env, H, W, ld3(dirs, l)));
}
void orc_ggx_specular(int n, int L, const float* normal, const float* pts2c, const float* pts2l /*[L,3]*/, const float* rough, const float* fresnel, float* out /*[n,L,3]*/) {
    for (int i = 0; i < n; i++) for (int l = 0; l < L; l++)
        st3(out, (size_t)i * L + l, ggx_specular(ld3(normal, i), ld3(pts2c, i), ld3(pts2l, l), ld3(rough, i), ld3(fresnel, i)));
}
void orc_dump_render(const int32_t* info, const float* aabb, const float* vert, const int32_t* tri, int n, int L, const float* pos, const float* normal,
                     const float* albedo, const float* rough, const float* fresnel, const float* rays_d, const float* light_dirs, const float* light_w,
                     const float* light_rgb, int equal_areas, int clamp_rgb, float* out_rgb, float* out_diff, float* out_spec) {
    Bvh B = {info, aabb, vert, tri};
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; i++) {
        f3 c, d, s;
        dump_render_point(B, ld3(pos, i), ld3(normal, i), ld3(albedo, i), ld3(rough, i), ld3(fresnel, i), ld3(rays_d, i), L, light_dirs, light_w, light_rgb, equal_areas != 0, c, d, s);
        if (clamp_rgb) c = mk3(clampf(c.x, 0.f, 1.f), clampf(c.y, 0.f, 1.f), clampf(c.z, 0.f, 1.f));   // dump_render :129
        st3(out_rgb, i, c); st3(out_diff, i, d); st3(out_spec, i, s);
    }
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
