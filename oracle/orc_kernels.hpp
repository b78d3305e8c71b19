// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header). PARITY UNPINNED.
// Per-pixel ReSTIR / shading / path-tracing / denoise kernels restated from
// nerf/ScreenSpaceReSTIR/{InitialResampling,TemporalResampling,SpatialResampling,EvaluateFinalSamples,
// FinalShading,EAWDenoise}.slang and utils/res.slang.
#pragma once
#include "orc_math.hpp"
#include "orc_bvh.hpp"
#include "orc_light.hpp"
#include "orc_brdf.hpp"

namespace orc {

// ReSTIR constants hard-coded by load_m_for_restir (renderer_restir.py:151-181) and in-shader defines.
struct Config {
    int light_tile_count = 128, light_tile_size = 1024, screen_tile_size = 8;
    int initial_light_samples = 32, initial_brdf_samples = 1;
    int max_history = 20, neighbor_offset_count = 8192, neighbor_count = 5;
    float gather_radius = 30.f;
    int max_bounce = 2;           // FinalShading.slang:7 MAX_Bounce
    float vis_near = 0.01f;       // VIS_near
};

struct Reservoirs { float* light_data; float* light_pdf; int32_t* M; float* weight; };  // [N,3],[N],[N],[N]

struct GBuf {  // per-pixel inputs, all [N,*] row-major
    int fx, fy;
    const float* occ;           // [N]
    const float* pos;           // [N,3]
    const float* normal_depth;  // [N,4]
    const float* brdf;          // [N,3] (diffuseWeight, specularWeight, ggxAlpha)  renderer_restir.py:281-287
    const float* ray_dir;       // [N,3]
};

struct RisState {  // res.slang:21-32
    f3 light_data; float inv_pdf; float weightSum, M, weight, canonicalWeight;
};
static inline RisState empty_ris() { RisState s; s.light_data = mk3(0.f); s.inv_pdf = 0.f; s.weightSum = 0.f; s.M = 0.f; s.weight = 0.f; s.canonicalWeight = 0.f; return s; }
struct Res { f3 light_data; float light_pdf; int M; float weight; };

static inline f3 ld3(const float* p, size_t i) { return mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
static inline void st3(float* p, size_t i, f3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
static inline Res load_res(const Reservoirs& R, size_t i) { Res r; r.light_data = ld3(R.light_data, i); r.light_pdf = R.light_pdf[i]; r.M = R.M[i]; r.weight = R.weight[i]; return r; }
static inline void store_zero(const Reservoirs& R, size_t i) { st3(R.light_data, i, mk3(0.f)); R.light_pdf[i] = 0.f; R.M[i] = 0; R.weight[i] = 0.f; }
static inline void store_ris(const Reservoirs& R, size_t i, const RisState& s) {
    st3(R.light_data, i, s.light_data); R.light_pdf[i] = s.inv_pdf; R.M[i] = (int)s.M; R.weight[i] = s.weight;
    if (std::isinf(s.weight) || std::isnan(s.weight)) store_zero(R, i);
}

static inline bool shadow_ray(const Bvh& B, f3 pos, f3 dir, float vis_near, TraceCounters* tc) {
    f3 o = pos + vis_near * dir;
    return bvh_hit(B, o, dir, 0.f, 1e7f, false, tc).hit;
}

// ------------------------------------------------------------ process_InitialResampling_  InitialResampling.slang:151-295
static inline void initial_pixel(const Config& C, const Bvh& B, const Env& E, const GBuf& G, const Reservoirs& R,
                                 const float* tile_data, const float* tile_pdf, uint32_t frameIndex, int x, int y, TraceCounters* tc) {
    size_t pi = (size_t)y * G.fx + x;
    if (G.occ[pi] < 0.1f) { store_zero(R, pi); return; }
    uint32_t tileSg = seed_generator((uint32_t)x / C.screen_tile_size, (uint32_t)y / C.screen_tile_size, frameIndex);
    uint32_t tileIndex = std::min((uint32_t)(next1d(tileSg) * C.light_tile_count), (uint32_t)C.light_tile_count - 1);
    uint32_t tileOffset = tileIndex * C.light_tile_size;
    uint32_t sg = seed_generator((uint32_t)x, (uint32_t)y, frameIndex);
    uint32_t stride = (C.light_tile_size + C.initial_light_samples - 1) / C.initial_light_samples;
    uint32_t offset = std::min((uint32_t)(next1d(sg) * stride), stride - 1);
    f3 n = mk3(G.normal_depth[4 * pi], G.normal_depth[4 * pi + 1], G.normal_depth[4 * pi + 2]);
    f3 rd = ld3(G.ray_dir, pi), br = ld3(G.brdf, pi);
    const float ratio = (float)C.initial_brdf_samples / (float)(C.initial_light_samples + C.initial_brdf_samples);
    RisState s = empty_ris();
    for (uint32_t i = 0; i < (uint32_t)C.initial_light_samples; ++i) {
        uint32_t index = tileOffset + (offset + i * stride) % C.light_tile_size;
        f3 ld = ld3(tile_data, index); float lpdf = tile_pdf[index];
        f3 em, ldir;
        get_light_info(E, mk2(ld.y, ld.z), em, ldir);
        float targetPdf = rt::target(em, ldir, n, rd, br);
        float brdfPdf = rt::eval_pdf_brdf(ldir, -rd, n, br.z, br.x, br.y);
        float sourcePdf = lerpf(lpdf, brdfPdf, ratio);  // evalInitialSamplePdf res.slang:79-91
        // streamingResampleStep res.slang:93-113
        float w = targetPdf / sourcePdf;
        s.weightSum += w; s.M += 1.f;
        if (next1d(sg) * s.weightSum < w) { s.light_data = ld; s.inv_pdf = lpdf; s.weight = targetPdf; }
    }
    for (int i = 0; i < C.initial_brdf_samples; ++i) {
        f3 ld = mk3(0.f); float lpdf = 0.f;
        f3 dir;
        float xa = next1d(sg), xb = next1d(sg), xc = next1d(sg);
        if (rt::sample_brdf(mk3(xa, xb, xc), dir, -rd, n, br.z, br.x, br.y)) {
            lpdf = pdf_li(E, dir);
            f2 o = oct_encode(dir);
            ld = mk3(1.0f, o.x, o.y);
        }
        if (ld.x < 0.1f) { s.M += 1.f; continue; }
        f3 em = env_le(ngp_dir(dir), E.tex, E.W, E.H);
        float targetPdf = rt::target(em, dir, n, rd, br);
        float brdfPdf = rt::eval_pdf_brdf(dir, -rd, n, br.z, br.x, br.y);
        float sourcePdf = lerpf(lpdf, brdfPdf, ratio);
        float w = targetPdf / sourcePdf;
        s.weightSum += w; s.M += 1.f;
        if (next1d(sg) * s.weightSum < w) { s.light_data = ld; s.inv_pdf = lpdf; s.weight = targetPdf; }
    }
    if (s.light_data.x > 0.1f) {
        f3 ldir = oct_decode(mk2(s.light_data.y, s.light_data.z));
        if (shadow_ray(B, ld3(G.pos, pi), ldir, C.vis_near, tc)) s = empty_ris();
    }
    s.weight = s.weight > 0.f ? (s.weightSum / s.M) / s.weight : 0.f;
    s.M = 1.f;
    store_ris(R, pi, s);
}

// ------------------------------------------------------------ process_TemporalResampling  TemporalResampling.slang:23-135
struct PrevGBuf { const float* occ; const float* normal_depth; const float* brdf; const float* ray_dir; };
static inline void temporal_pixel(const Config& C, const Env& E, const GBuf& G, const PrevGBuf& P, const Reservoirs& R,
                                  const Reservoirs& prevR, const float* motion, uint32_t frameIndex, int x, int y) {
    size_t pi = (size_t)y * G.fx + x;
    if (G.occ[pi] < 0.1f) return;
    uint32_t sg = seed_generator((uint32_t)x, (uint32_t)y, frameIndex);
    float mvx = motion ? motion[2 * pi] : 0.f, mvy = motion ? motion[2 * pi + 1] : 0.f;
    float jx = next1d(sg), jy = next1d(sg);
    int ppx = (int)(((float)(uint32_t)x + mvx * (float)(uint32_t)G.fx) + (jx * 1.f - 0.f));
    int ppy = (int)(((float)(uint32_t)y + mvy * (float)(uint32_t)G.fy) + (jy * 1.f - 0.f));
    if (ppx >= G.fx || ppx < 0 || ppy >= G.fy || ppy < 0) return;
    size_t qi = (size_t)ppy * G.fx + ppx;
    if (P.occ[qi] < 0.1f) return;
    f3 n = mk3(G.normal_depth[4 * pi], G.normal_depth[4 * pi + 1], G.normal_depth[4 * pi + 2]);
    float depth = G.normal_depth[4 * pi + 3];
    f3 rd = ld3(G.ray_dir, pi), br = ld3(G.brdf, pi);
    f3 pn = mk3(P.normal_depth[4 * qi], P.normal_depth[4 * qi + 1], P.normal_depth[4 * qi + 2]);
    float pdepth = P.normal_depth[4 * qi + 3];
    f3 prd = ld3(P.ray_dir, qi), pbr = ld3(P.brdf, qi);
    Res cur = load_res(R, pi), prev = load_res(prevR, qi);
    prev.M = std::min(prev.M, cur.M * C.max_history);
    RisState s = empty_ris();
    // isValidNeighbor res.slang:63-68
    if (!(dot(n, pn) >= 0.5f && fabsf(depth - pdepth) <= 0.1f * depth)) return;
    f3 em, ldir;
    get_light_info(E, mk2(cur.light_data.y, cur.light_data.z), em, ldir);
    float targetPdf = rt::target(em, ldir, n, rd, br);
    {   // streamingResampleStep(reservoir) res.slang:116-134
        float w = targetPdf * cur.weight * cur.M;
        s.weightSum += w; s.M += cur.M;
        if (next1d(sg) * s.weightSum < w) { s.light_data = cur.light_data; s.inv_pdf = cur.light_pdf; s.weight = targetPdf; }
    }
    f3 pem, pldir;
    get_light_info(E, mk2(prev.light_data.y, prev.light_data.z), pem, pldir);
    float preTarget = rt::target(pem, pldir, n, rd, br);
    bool usedPrev;
    {
        float w = preTarget * prev.weight * prev.M;
        s.weightSum += w; s.M += prev.M;
        usedPrev = next1d(sg) * s.weightSum < w;
        if (usedPrev) { s.light_data = prev.light_data; s.inv_pdf = prev.light_pdf; s.weight = preTarget; }
    }
    f3 sem, sdir;
    get_light_info(E, mk2(s.light_data.y, s.light_data.z), sem, sdir);
    float currentPdf = rt::target(sem, sdir, n, rd, br);
    float prevPdf = rt::target(sem, sdir, pn, prd, pbr);
    float normalization = (usedPrev ? prevPdf : currentPdf) / (cur.M * currentPdf + prev.M * prevPdf);
    s.weight = s.weight > 0.f ? (s.weightSum * normalization) / s.weight : 0.f;
    store_ris(R, pi, s);
}

// ------------------------------------------------------------ process_SpatialResampling_  SpatialResampling.slang:178-322
static inline float m_factor(float q0, float q1) { return q0 == 0.f ? 1.f : clampf(mrf_pow2k(fminf(q1 / q0, 1.f), 3), 0.f, 1.f); }
static inline float pairwise_mis(float q0, float q1, float N0, float N1) { return (q1 == 0.f) ? 0.f : (N0 * q0) / (q0 * N0 + q1 * N1); }

// Test hook (tests/test_oracle_invariants.py): -1 = the reference's behaviour. 0 / 1: every spatial shadow ray aimed at the light sample of a reservoir with
// weight == 0 (an emptied one) is not traced and reports "free" / "occluded". The claim under test: such a ray's answer cannot reach the output — it only ever
// scales a term that is multiplied by that weight (w = candAtOther * nb.weight * m0; w_final = curTarget * cur.weight * canonicalWeight), res.slang:173-232.
static int g_dead_ray_override = -1;
static long long g_dead_ray_count = 0;        // rays the hook answered (0 / 1); mode 2 = control: EVERY spatial shadow ray reports "occluded" (the frame must change)
static inline bool dead_ray_answer(bool dead, bool& hit) {
    if (g_dead_ray_override < 0 || !(dead || g_dead_ray_override == 2)) return false;
    hit = g_dead_ray_override != 0;
    #pragma omp atomic
    g_dead_ray_count++;
    return true;
}
static inline void spatial_pixel(const Config& C, const Bvh& B, const Env& E, const GBuf& G, const Reservoirs& R,
                                 const Reservoirs& prevR, const float* neighborOffsets, uint32_t frameIndex, int x, int y, TraceCounters* tc) {
    size_t pi = (size_t)y * G.fx + x;
    if (G.occ[pi] < 0.1f) { store_zero(R, pi); return; }
    uint32_t sg = seed_generator((uint32_t)x, (uint32_t)y, frameIndex);
    f3 n = mk3(G.normal_depth[4 * pi], G.normal_depth[4 * pi + 1], G.normal_depth[4 * pi + 2]);
    float depth = G.normal_depth[4 * pi + 3];
    f3 rd = ld3(G.ray_dir, pi), br = ld3(G.brdf, pi);
    RisState s = empty_ris();
    const uint32_t startIndex = (uint32_t)(next1d(sg) * C.neighbor_offset_count);
    Res cur = load_res(prevR, pi);
    f3 cem, cdir;
    get_light_info(E, mk2(cur.light_data.y, cur.light_data.z), cem, cdir);
    float curTarget = rt::target(cem, cdir, n, rd, br);
    f3 cpos = ld3(G.pos, pi);
    s.canonicalWeight = 1.f;
    uint32_t validNeighbors = 1;
    const uint32_t k = (uint32_t)C.neighbor_count;
    for (uint32_t i = 0; i < k; ++i) {
        uint32_t ni = (startIndex + i) & (uint32_t)(C.neighbor_offset_count - 1);
        int nx = x + (int)(neighborOffsets[2 * ni] * C.gather_radius);
        int ny = y + (int)(neighborOffsets[2 * ni + 1] * C.gather_radius);
        if (!(nx >= 0 && ny >= 0 && nx < G.fx && ny < G.fy)) continue;
        size_t qi = (size_t)ny * G.fx + nx;
        f3 nn = mk3(G.normal_depth[4 * qi], G.normal_depth[4 * qi + 1], G.normal_depth[4 * qi + 2]);
        float ndepth = G.normal_depth[4 * qi + 3];
        if (!(dot(n, nn) >= 0.5f && fabsf(depth - ndepth) <= 0.1f * depth)) continue;
        Res nb = load_res(prevR, qi);
        if (nb.M == 0) continue;
        f3 nrd = ld3(G.ray_dir, qi), nbr = ld3(G.brdf, qi);
        if (G.occ[qi] < 0.1f) continue;
        ++validNeighbors;
        f3 nem, ndir;
        get_light_info(E, mk2(nb.light_data.y, nb.light_data.z), nem, ndir);
        f3 npos = ld3(G.pos, qi);
        bool canonical_hit, candidate_hit;
        if (!dead_ray_answer(nb.weight == 0.f, canonical_hit)) canonical_hit = shadow_ray(B, cpos, ndir, C.vis_near, tc);
        if (!dead_ray_answer(cur.weight == 0.f, candidate_hit)) candidate_hit = shadow_ray(B, npos, cdir, C.vis_near, tc);
        float canonicalVis = canonical_hit ? 0.f : 1.f, candidateVis = candidate_hit ? 0.f : 1.f;
        // streamingResampleStepMisUnbiased res.slang:173-213
        float candTarget = rt::target(nem, ndir, nn, nrd, nbr);
        float candAtOther = rt::target(nem, ndir, n, rd, br);
        float canonAtOther = rt::target(cem, cdir, nn, nrd, nbr);
        candAtOther *= canonicalVis;
        canonAtOther *= candidateVis;
        float N0 = (float)((uint32_t)nb.M * k), N1 = (float)cur.M;
        float m0 = pairwise_mis(candTarget, candAtOther, N0, N1);
        float m1 = 1.f - pairwise_mis(canonAtOther, curTarget, N0, N1);
        float w = candAtOther * nb.weight * m0;
        s.M += nb.M * fminf(m_factor(candTarget, candAtOther), m_factor(canonAtOther, curTarget));
        s.weightSum += w;
        s.canonicalWeight += m1;
        if (next1d(sg) * s.weightSum < w) { s.light_data = nb.light_data; s.inv_pdf = nb.light_pdf; s.weight = candAtOther; }
    }
    {   // streamingResampleFinalizeMis res.slang:215-232
        float w = curTarget * cur.weight * s.canonicalWeight;
        s.M += cur.M; s.weightSum += w;
        if (next1d(sg) * s.weightSum < w) { s.light_data = cur.light_data; s.inv_pdf = cur.light_pdf; s.weight = curTarget; }
    }
    s.M = (float)cur.M;
    s.weight = s.weight > 0.f ? (s.weightSum / validNeighbors) / s.weight : 0.f;
    store_ris(R, pi, s);
}

// ------------------------------------------------------------ EvaluateFinalSamples.slang:84-188
static inline void final_vis_pixel(const Config& C, const Bvh& B, const GBuf& G, const Reservoirs& R, float* vis, int x, int y, TraceCounters* tc) {
    size_t pi = (size_t)y * G.fx + x;
    f3 ld = ld3(R.light_data, pi);
    vis[pi] = 1.0f;
    if (ld.x > 0.1f) {
        f3 ldir = oct_decode(mk2(ld.y, ld.z));
        vis[pi] = shadow_ray(B, ld3(G.pos, pi), ldir, C.vis_near, tc) ? 0.0f : 1.0f;
    }
}
static inline void eval_final_pixel(const Env& E, const Reservoirs& R, const float* vis, float* fdir, float* fdist, float* fLi, size_t pi) {
    Res cur = load_res(R, pi);
    st3(fdir, pi, mk3(0.f)); fdist[pi] = 0.f;
    f3 Li = mk3(0.f);
    if (cur.light_data.x > 0.1f) {
        f3 em, ldir;
        get_light_info(E, mk2(cur.light_data.y, cur.light_data.z), em, ldir);
        if (vis[pi] > 0.f) { st3(fdir, pi, ldir); fdist[pi] = 1e6f; Li = cur.weight * em; }
    }
    st3(fLi, pi, Li);
}

// ------------------------------------------------------------ process_FinalShading  FinalShading.slang:14-109
static inline void final_shading_pixel(const Env& E, const float* occ, const float* normal, const float* ray_dir, const float* kd,
                                       const float* rs, const float* fdir, const float* fdist, const float* fLi,
                                       float* color, float* diff_light, float* spec_light, size_t pi) {
    f3 n = ld3(normal, pi), rd = ld3(ray_dir, pi), diffuse = ld3(kd, pi);
    float rough = rs[2 * pi], metallic = rs[2 * pi + 1];
    f3 c = mk3(0.f), ldiff = mk3(0.f), lspec = mk3(0.f);
    if (occ[pi] > 0.1f) {
        f3 dir = ld3(fdir, pi); float distance = fdist[pi]; f3 Li = ld3(fLi, pi);
        f3 dv = mk3(0.f), sv = mk3(0.f);
        if (distance > 0.f) {
            sh::Frame fr = sh::create_frame(n);
            f3 wi = sh::to_local(fr, -rd), wo = sh::to_local(fr, dir);
            sh::Lobes L = sh::lobes(diffuse, rough, metallic, rd, n);
            if (L.pD > 0.f) dv = sh::diffuse_light(wi, wo) * Li;
            if (L.pS > 0.f) sv = sh::specular_eval(wi, wo, L.specular, L.alpha) * Li;
        }
        c += diffuse * (1.0f - metallic) * dv + sv;
        ldiff += dv; lspec += sv;
    } else {
        c = env_le(ngp_dir(rd), E.tex, E.W, E.H);
    }
    st3(color, pi, c); st3(diff_light, pi, ldiff); st3(spec_light, pi, lspec);
}

// ------------------------------------------------------------ path tracing
struct PathBufs {  // one vertex generation
    const float* occ; const float* pos; const float* normal; const float* ray_dir;  // inputs [N,*]
    const float* kd; const float* rs;                                               // [N,3],[N,2]
    float* prd;                                                                      // [N,5]
    float* new_pos; float* new_ray_d; float* new_occ; float* new_normal;             // outputs
};

// shared tail of process_new_dir_for_pt (:229-262) and process_path_tracing_divided_no_grad (:921-983)
static inline void next_bounce(const Config& C, const Bvh& B, const PathBufs& P, size_t pi, uint32_t bounce_count, const sh::Frame& fr,
                               const sh::Lobes& L, f3 wiLocal, f3 diffuse_col, f3 surf_pos, f3 throughput, uint32_t& sg, TraceCounters* tc) {
    f3 out_dir; float out_pdf; f3 out_weight; uint32_t sampledSpecular = 0;
    bool valid = sh::falcor_sample(L.pD, L.pS, wiLocal, out_dir, out_pdf, sampledSpecular, out_weight, sg, L.alpha, L.specular, diffuse_col, true);
    if (!valid) return;
    if (is_black(out_weight) || out_pdf == 0.f) { P.prd[5 * pi + 4] = 1.f; return; }
    if (bounce_count + 1 <= (uint32_t)C.max_bounce) {
        out_dir = normalize(sh::to_global(fr, out_dir));
        f3 o = surf_pos + C.vis_near * out_dir;
        HitResult h = bvh_hit(B, o, out_dir, 0.f, 1e7f, true, tc);
        float specularBounce = (float)sampledSpecular;
        throughput *= out_weight;
        P.prd[5 * pi] = throughput.x; P.prd[5 * pi + 1] = throughput.y; P.prd[5 * pi + 2] = throughput.z;
        P.prd[5 * pi + 3] = specularBounce;
        st3(P.new_ray_d, pi, out_dir);
        if (h.hit) {
            P.prd[5 * pi + 4] = 0.f;
            st3(P.new_pos, pi, h.pos); st3(P.new_normal, pi, h.normal); P.new_occ[pi] = 1.f;
        } else if (specularBounce > 0.f) P.prd[5 * pi + 4] = 0.f;
    }
}

// process_new_dir_for_pt  FinalShading.slang:113-265
static inline void new_dir_pixel(const Config& C, const Bvh& B, const PathBufs& P, int fx, uint32_t frameIndex, uint32_t bounce_count, int x, int y, TraceCounters* tc) {
    size_t pi = (size_t)y * fx + x;
    f3 thr = mk3(P.prd[5 * pi], P.prd[5 * pi + 1], P.prd[5 * pi + 2]);
    float is_stop = P.prd[5 * pi + 4];
    P.new_occ[pi] = 0.f; P.prd[5 * pi + 4] = 1.f;
    if (bounce_count == 0) { thr = mk3(1.0f); is_stop = 0.f; P.prd[5 * pi] = 1.f; P.prd[5 * pi + 1] = 1.f; P.prd[5 * pi + 2] = 1.f; P.prd[5 * pi + 3] = 0.f; }
    if (is_stop > 0.f) return;
    f3 n = ld3(P.normal, pi), rd = ld3(P.ray_dir, pi), sp = ld3(P.pos, pi), diffuse = ld3(P.kd, pi);
    float rough = P.rs[2 * pi], metallic = P.rs[2 * pi + 1];
    uint32_t sg = seed_generator((uint32_t)x, (uint32_t)y, frameIndex);
    if (P.occ[pi] > 0.1f) {
        sh::Lobes L = sh::lobes(diffuse, rough, metallic, rd, n);
        sh::Frame fr = sh::create_frame(n);
        f3 wi = sh::to_local(fr, -rd);
        next_bounce(C, B, P, pi, bounce_count, fr, L, wi, diffuse * (1.0f - metallic), sp, thr, sg, tc);
    }
}

// process_path_tracing_divided_no_grad  FinalShading.slang:641-1009
static inline void bounce_pixel(const Config& C, const Bvh& B, const Env& E, const PathBufs& P, int fx, uint32_t frameIndex, uint32_t bounce_count,
                                float* color, float* diff_color, float* spec_color, int x, int y, TraceCounters* tc) {
    size_t pi = (size_t)y * fx + x;
    f3 thr = mk3(P.prd[5 * pi], P.prd[5 * pi + 1], P.prd[5 * pi + 2]);
    float specularBounce = P.prd[5 * pi + 3];
    float is_stop = P.prd[5 * pi + 4];
    P.new_occ[pi] = 0.f; P.prd[5 * pi + 4] = 1.f;
    if (bounce_count == 0) { thr = mk3(1.0f); specularBounce = 0.f; is_stop = 0.f; P.prd[5 * pi] = 1.f; P.prd[5 * pi + 1] = 1.f; P.prd[5 * pi + 2] = 1.f; P.prd[5 * pi + 3] = 0.f; }
    if (is_stop > 0.f) { st3(color, pi, mk3(0.f)); st3(diff_color, pi, mk3(0.f)); st3(spec_color, pi, mk3(0.f)); return; }
    f3 n = ld3(P.normal, pi), rd = ld3(P.ray_dir, pi), sp = ld3(P.pos, pi), diffuse = ld3(P.kd, pi);
    float rough = P.rs[2 * pi], metallic = P.rs[2 * pi + 1];
    f3 cv = mk3(0.f), dcv = mk3(0.f), scv = mk3(0.f);
    uint32_t sg = seed_generator((uint32_t)x, (uint32_t)y, frameIndex);
    if (P.occ[pi] > 0.1f) {
        sh::Lobes L = sh::lobes(diffuse, rough, metallic, rd, n);
        float lightPdf = 0.0f, scatteringPdf = 0.0f;
        // NEE sample :745-760  (float2(sampleNext1D, sampleNext1D): source order)
        float r0 = next1d(sg), r1 = next1d(sg);
        f3 samp_dir = mk3(0.f); float samp_pdf = 0.f; f2 luv; f3 samp_weight = mk3(0.f); bool samp_valid = false;
        {
            f3 d; float p;
            if (sample_li(E, mk2(r0, r1), d, p, luv)) {
                samp_valid = true; samp_dir = d; samp_pdf = p;
                samp_weight = env_le(ngp_dir(d), E.tex, E.W, E.H) / p;
            }
        }
        sh::Frame fr = sh::create_frame(n);
        f3 wi = sh::to_local(fr, -rd);
        lightPdf = samp_pdf;
        f3 Li = samp_weight;
        f3 diffuse_col = diffuse * (1.0f - metallic);
        if (samp_valid && lightPdf > 0 && !is_black(Li)) {
            f3 diff_f = mk3(0.f), spec_f = mk3(0.f), total_f = mk3(0.f);
            f3 wo = sh::to_local(fr, samp_dir);
            if (!is_black(n)) {
                if (L.pD > 0.f) diff_f = sh::diffuse_light(wi, wo);
                if (L.pS > 0.f) spec_f = sh::specular_eval(wi, wo, L.specular, L.alpha);
                total_f = diffuse_col * diff_f + spec_f;
                diff_f = diffuse_col * diff_f;
                scatteringPdf = sh::falcor_eval_pdf(L.pD, L.pS, wi, wo, L.alpha);
            }
            if (!is_black(total_f)) {
                f3 ldir = normalize(samp_dir);
                bool hit = shadow_ray(B, sp, ldir, C.vis_near, tc);
                f3 tr = hit ? mk3(0.f) : mk3(1.f);
                Li *= tr;
                if (!is_black(Li)) {
                    float mis = lightPdf * lightPdf / (lightPdf * lightPdf + scatteringPdf * scatteringPdf);  // power_heuristic helperDi.slang:407-409
                    cv += thr * total_f * Li * mis;
                    dcv += thr * diff_f * Li * mis;
                    scv += thr * spec_f * Li * mis;
                }
            }
        }
        // BSDF sample with MIS :806-903
        uint32_t sampledSpecular = 0;
        if (!is_black(n)) {
            f3 m_wi; float m_pdf; f3 dummy;
            bool valid = sh::falcor_sample(L.pD, L.pS, wi, m_wi, m_pdf, sampledSpecular, dummy, sg, L.alpha, L.specular, diffuse_col, false);
            if (valid) {
                f3 bdw = mk3(1.0f), bsw = mk3(1.0f);
                if (L.pD > 0.f) bdw = sh::diffuse_light(wi, m_wi);
                if (L.pS > 0.f) bsw = sh::specular_eval(wi, m_wi, L.specular, L.alpha);
                f3 bw = diffuse_col * bdw + bsw;
                m_wi = sh::to_global(fr, m_wi);
                scatteringPdf = m_pdf;
                f3 f = bw / m_pdf, diff_f = diffuse_col * bdw / m_pdf, spec_f = bsw / m_pdf;
                f *= m_pdf; diff_f *= m_pdf; spec_f *= m_pdf;
                f3 safe_wi = normalize(m_wi);
                if (!is_black(f) && scatteringPdf > 0) {
                    float weight = 1.0f; bool lightZero = false;
                    if (sampledSpecular == 0) {
                        lightPdf = pdf_li(E, safe_wi);
                        if (lightPdf == 0.0f) lightZero = true;
                        weight = scatteringPdf * scatteringPdf / (scatteringPdf * scatteringPdf + lightPdf * lightPdf);
                    }
                    bool found = shadow_ray(B, sp, safe_wi, C.vis_near, tc);
                    f3 Tr = mk3(1.f);
                    Li = mk3(0.f);
                    if (!found) Li = env_le(ngp_dir(safe_wi), E.tex, E.W, E.H);
                    if (!is_black(Li) && !lightZero) {
                        cv += thr * f * Li * Tr * weight / scatteringPdf;
                        dcv += thr * diff_f * Li * Tr * weight / scatteringPdf;
                        scv += thr * spec_f * Li * Tr * weight / scatteringPdf;
                    }
                }
            }
        }
        next_bounce(C, B, P, pi, bounce_count, fr, L, wi, diffuse_col, sp, thr, sg, tc);
    } else {
        if (bounce_count == 0) cv += thr * env_le(ngp_dir(rd), E.tex, E.W, E.H);
        else if (specularBounce > 0.f) { f3 e = thr * env_le(ngp_dir(rd), E.tex, E.W, E.H); cv += e; scv += e; }
        P.prd[5 * pi + 4] = 1.f;
    }
    st3(color, pi, cv); st3(diff_color, pi, dcv); st3(spec_color, pi, scv);
}

// ------------------------------------------------------------ process_EAWDenoise(_no_di)  EAWDenoise.slang:50-302
static inline void eaw_pixel(int fx, int fy, int stepWidth, float c_phi, float n_phi, float p_phi, const float* occ, const float* color,
                             const float* normal, const float* pos, float* out, int x, int y) {
    size_t pi = (size_t)y * fx + x;
    if (occ[pi] < 0.1f) { st3(out, pi, ld3(color, pi)); return; }
    static const float kern1[5] = {1.f, 4.f, 6.f, 4.f, 1.f};  // B3 spline: kernel[i] = k[ix]*k[iy]/256 (values listed at :114-143)
    f3 nval = ld3(normal, pi), pval = ld3(pos, pi), cval = ld3(color, pi);
    f3 sum = mk3(0.f); float cum_w = 0.0f;
    for (int i = 0; i < 25; i++) {
        int ox = (i % 5) - 2, oy = (i / 5) - 2;
        int ux = x + (int)((float)ox * stepWidth), uy = y + (int)((float)oy * stepWidth);
        if (!(ux >= 0 && uy >= 0 && ux < fx && uy < fy)) continue;
        size_t qi = (size_t)uy * fx + ux;
        float kw = kern1[i % 5] * kern1[i / 5] / 256.0f;
        f3 ctmp = ld3(color, qi);
        f3 t = cval - ctmp;
        float dist2 = dot(t, t);
        float c_w = fminf(mrf_exp(-(dist2) / c_phi), 1.0f);
        f3 ntmp = ld3(normal, qi);
        t = nval - ntmp;
        dist2 = fmaxf(dot(t, t), 0.0f);
        float n_w = fminf(mrf_exp(-(dist2) / n_phi), 1.0f);
        f3 ptmp = ld3(pos, qi);
        t = pval - ptmp;
        dist2 = fmaxf(dot(t, t), 0.0f);
        float p_w = fminf(mrf_exp(-(dist2) / p_phi), 1.0f);
        float weight = c_w * n_w * p_w;
        sum += ctmp * weight * kw;
        cum_w += weight * kw;
    }
    st3(out, pi, sum / cum_w);
}

}  // namespace orc
