// shading.hip — final shading of the ReSTIR sample and the multi-bounce path-tracing vertices.
// process_FinalShading, process_new_dir_for_pt, process_path_tracing_divided_no_grad (FinalShading.slang:14-265, 641-1009)
// as generate -> trace -> resolve wavefront stages: a vertex emits up to two shadow rays (NEE, BSDF-MIS) into the
// any-hit queue and one continuation ray into the closest-hit queue; the contributions that depend on visibility are
// parked per pixel (18 floats) and committed by the resolve stage in the reference's summation order.
#ifndef MR_LEAN_FP
#define MR_LEAN_FP 1      // device_math.hpp: short division / square-root sequences, bit-identical to the compiler's for the renderer's operand range
#endif
#include "engine.hpp"
#include "device_math.hpp"
#include "device_light.hpp"
#include "device_brdf.hpp"

namespace mr {

int trace_any_queue_counted(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, int32_t* hit,
                            unsigned long long* stats, hipStream_t s, int reference_order);
int trace_closest_queue_counted(const mirres_bvh* bvh, const Ray* rays, const uint32_t* d_count, size_t capacity, HitRec* out,
                                unsigned long long* stats, hipStream_t s);

#define MR_BLOCK 256
#ifndef MR_GEN_BLOCK
#define MR_GEN_BLOCK 512    // ray-generating kernels: one queue atomic per block
#endif
#ifndef MR_BGEN_BLOCK
#define MR_BGEN_BLOCK MR_GEN_BLOCK
#endif

MR_DEV void put_ray(Ray* q, uint32_t slot, v3 pos, v3 dir, float vis_near) {
    v3 o = pos + vis_near * dir;
    float4 a, b;
    a.x = o.x; a.y = o.y; a.z = o.z; a.w = 0.f;
    b.x = dir.x; b.y = dir.y; b.z = dir.z; b.w = 1e7f;
    reinterpret_cast<float4*>(q + slot)[0] = a; reinterpret_cast<float4*>(q + slot)[1] = b;
}

// ---------------------------------------------------------------- process_FinalShading (FinalShading.slang:14-109)
template <bool ACC>
__global__ void __launch_bounds__(MR_BLOCK) k_final_shading(EnvD E, const float* __restrict__ occ, const float* __restrict__ normal,
                                                            const float* __restrict__ ray_dir, const float* __restrict__ kd, const float* __restrict__ rm,
                                                            const float* __restrict__ fdir, const float* __restrict__ fdist, const float* __restrict__ fLi,
                                                            int N, float* __restrict__ color, float* __restrict__ diff_light, float* __restrict__ spec_light) {
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= N) return;
    const v3 n = ld3(normal, pi), rd = ld3(ray_dir, pi), diffuse = ld3(kd, pi);
    const float rough = rm[2 * (size_t)pi], metallic = rm[2 * (size_t)pi + 1];
    v3 c = V3(0.f), ldiff = V3(0.f), lspec = V3(0.f);
    if (occ[pi] > 0.1f) {
        v3 dv = V3(0.f), sv = V3(0.f);
        if (fdist[pi] > 0.f) {
            const v3 dir = ld3(fdir, pi), Li = ld3(fLi, pi);
            shade::Frame fr = shade::create_frame(n);
            v3 wi = shade::to_local(fr, -rd), wo = shade::to_local(fr, dir);
            shade::Lobes L = shade::lobes(diffuse, rough, metallic, rd, n);
            if (L.pD > 0.f) dv = shade::diffuse_light(wi, wo) * Li;
            if (L.pS > 0.f) sv = shade::specular_eval(wi, wo, L.specular, L.alpha) * Li;
        }
        c = diffuse * (1.0f - metallic) * dv + sv;
        ldiff = dv; lspec = sv;
    } else {
        c = env_le(ngp_dir(rd), E.tex, E.W, E.H);
    }
    if (ACC) {  // fused accumulation (renderer_restir.py:466-468)
        st3(color, pi, ld3(color, pi) + c); st3(diff_light, pi, ld3(diff_light, pi) + ldiff); st3(spec_light, pi, ld3(spec_light, pi) + lspec);
    } else { st3(color, pi, c); st3(diff_light, pi, ldiff); st3(spec_light, pi, lspec); }
}

// ---------------------------------------------------------------- continuation ray: tail shared by new_dir (:229-262) and the bounce kernel (:921-983)
struct Vertex { v3 n, rd, pos, diffuse; float rough, metallic; };
MR_DEV Vertex load_vertex(const mirres_path_t& P, size_t pi) {
    Vertex v; v.n = ld3(P.normal, pi); v.rd = ld3(P.ray_dir, pi); v.pos = ld3(P.pos, pi); v.diffuse = ld3(P.kd, pi);
    v.rough = P.rough_metal[2 * pi]; v.metallic = P.rough_metal[2 * pi + 1];
    return v;
}
// returns true when a continuation ray must be traced (dir written to out_dir)
MR_DEV bool next_bounce_gen(const mirres_path_t& P, size_t pi, int max_bounce, uint32_t bounce_count, const shade::Frame& fr, const shade::Lobes& L,
                            v3 wiLocal, v3 diffuse_col, v3 throughput, uint32_t& sg, v3& out_dir) {
    v3 wo; float out_pdf; v3 out_weight; uint32_t sampledSpecular = 0;
    bool valid = shade::falcor_sample<true>(L.pD, L.pS, wiLocal, wo, out_pdf, sampledSpecular, out_weight, sg, L.alpha, L.specular, diffuse_col);
    if (!valid) return false;
    if (is_black(out_weight) || out_pdf == 0.f) { P.prd[5 * pi + 4] = 1.f; return false; }
    if (bounce_count + 1 <= (uint32_t)max_bounce) {
        out_dir = normalize(shade::to_global(fr, wo));
        throughput = throughput * out_weight;
        P.prd[5 * pi] = throughput.x; P.prd[5 * pi + 1] = throughput.y; P.prd[5 * pi + 2] = throughput.z;
        P.prd[5 * pi + 3] = (float)sampledSpecular;
        st3(P.new_ray_d, pi, out_dir);
        return true;
    }
    return false;
}
// returns whether the path goes on (the slot has work at the next vertex)
MR_DEV bool next_bounce_resolve(const mirres_path_t& P, size_t pi, const HitRec* __restrict__ rec, int slot) {
    const float4 a = reinterpret_cast<const float4*>(rec + slot)[0], b = reinterpret_cast<const float4*>(rec + slot)[1];
    if (__float_as_int(a.w)) {
        P.prd[5 * pi + 4] = 0.f;
        st3(P.new_pos, pi, V3(a.x, a.y, a.z)); st3(P.new_normal, pi, V3(b.x, b.y, b.z)); P.new_occ[pi] = 1.f;
        return true;
    } else if (P.prd[5 * pi + 3] > 0.f) { P.prd[5 * pi + 4] = 0.f; return true; }  // specular bounce can pick up the env map at the next vertex
    return false;
}
// thread -> slot: every slot (list == NULL) or entry `t` of a live list (an invalid slot beyond its device-side count)
MR_DEV int live_slot(const int32_t* __restrict__ list, const uint32_t* __restrict__ count, int t, int none) {
    if (!list) return t;
    return (uint32_t)t < *count ? list[t] : none;
}

// ---------------------------------------------------------------- process_new_dir_for_pt (FinalShading.slang:113-265)
// Sample slots: thread v = k * N + pixel works on sample k of a K-sample batch (NV = K * N; K = 1 for the stepwise ABI). The vertex inputs of
// new_dir are the G-buffer (one per pixel, shared by the K samples); the outputs and everything in the bounce kernels are per slot. The RNG
// stream of a slot is the one its sample would get in a sample-by-sample loop: frameIndex + 20 * k (mTotalRISPasses), minus one for the
// frame's very first sample, which has no temporal pass before it (renderer_restir.py:341-356).
MR_DEV uint32_t slot_frame(uint32_t frameIndex, int k, int first_is_zero) { return frameIndex + 20u * (uint32_t)k - ((first_is_zero && k == 0) ? 1u : 0u); }
__global__ void __launch_bounds__(MR_GEN_BLOCK) k_new_dir_gen(mirres_path_t P, int max_bounce, float vis_near, uint32_t frameIndex, uint32_t bounce_count, int fx,
                                                          int N, int NV, int first_is_zero, int y_off, Ray* __restrict__ q, uint32_t* __restrict__ q_count, int32_t* __restrict__ slot_out) {
    const int v_ = blockIdx.x * blockDim.x + threadIdx.x;
    bool want = false; v3 rp = V3(0.f), rdir = V3(0.f);
    if (v_ < NV) {
        const int k = v_ / N, pi = v_ - k * N;
        const size_t sv = (size_t)v_;
        v3 thr = V3(P.prd[5 * sv], P.prd[5 * sv + 1], P.prd[5 * sv + 2]);
        float is_stop = P.prd[5 * sv + 4];
        P.new_occ[sv] = 0.f; P.prd[5 * sv + 4] = 1.f;
        if (bounce_count == 0) { thr = V3(1.0f); is_stop = 0.f; P.prd[5 * sv] = 1.f; P.prd[5 * sv + 1] = 1.f; P.prd[5 * sv + 2] = 1.f; P.prd[5 * sv + 3] = 0.f; }
        if (!(is_stop > 0.f) && P.occ[pi] > 0.1f) {
            Vertex v = load_vertex(P, pi);
            uint32_t sg = seed_generator((uint32_t)(pi % fx), (uint32_t)(pi / fx + y_off), slot_frame(frameIndex, k, first_is_zero));
            shade::Lobes L = shade::lobes(v.diffuse, v.rough, v.metallic, v.rd, v.n);
            shade::Frame fr = shade::create_frame(v.n);
            v3 wi = shade::to_local(fr, -v.rd);
            want = next_bounce_gen(P, sv, max_bounce, bounce_count, fr, L, wi, v.diffuse * (1.0f - v.metallic), thr, sg, rdir);
            rp = v.pos;
        }
    }
    uint32_t slot = block_append(q_count, want);
    if (want) put_ray(q, slot, rp, rdir, vis_near);
    if (v_ < NV) slot_out[v_] = want ? (int32_t)slot : -1;
}
// MR_RES_PER consecutive slots per thread: one queue atomic per 2048 slots (a queue-head word takes ~88 atomics per microsecond; with one slot per thread
// the 320 k appends of a batch were what this kernel spent its time on: 1.86 ms against 0.81 ms without the list)
#define MR_RES_PER 8
__global__ void __launch_bounds__(MR_BLOCK) k_new_dir_resolve(mirres_path_t P, int N, const int32_t* __restrict__ slot, const HitRec* __restrict__ rec,
                                                              int32_t* __restrict__ live_out, uint32_t* __restrict__ live_count) {
    const int base = blockIdx.x * (MR_BLOCK * MR_RES_PER) + threadIdx.x;      // slots base + j * MR_BLOCK: every load of the wave is one contiguous run
    uint32_t alive = 0;
#pragma unroll
    for (int j = 0; j < MR_RES_PER; j++) {
        const int pi = base + j * MR_BLOCK;
        if (pi < N) {
            const int s = slot[pi];
            if (s >= 0 && next_bounce_resolve(P, pi, rec, s)) alive |= 1u << j;
        }
    }
    if (live_out) {   // the slots the first indirect vertex has work for (every thread of the block takes part in the append)
        uint32_t o = block_append(live_count, alive != 0, __popc(alive));
#pragma unroll
        for (int j = 0; j < MR_RES_PER; j++) if (alive & (1u << j)) live_out[o++] = base + j * MR_BLOCK;
    }
}

// ---------------------------------------------------------------- process_path_tracing_divided_no_grad (FinalShading.slang:641-1009)
#ifdef MR_BGEN_WAVES
__attribute__((amdgpu_waves_per_eu(MR_BGEN_WAVES, MR_BGEN_WAVES)))
#endif
__global__ void __launch_bounds__(MR_BGEN_BLOCK) k_bounce_gen(mirres_path_t P, EnvD E, int max_bounce, float vis_near, uint32_t frameIndex, uint32_t bounce_count,
                                                         int fx, int N, int NV, int first_is_zero, int y_off, int sparse, float* __restrict__ color, float* __restrict__ diff_color, float* __restrict__ spec_color,
                                                         Ray* __restrict__ qa, uint32_t* __restrict__ qa_count, Ray* __restrict__ qc, uint32_t* __restrict__ qc_count,
                                                         int32_t* __restrict__ slot_a, uint32_t* __restrict__ mask_out, int32_t* __restrict__ slot_c,
                                                         float* __restrict__ pend, const int32_t* __restrict__ live_in, const uint32_t* __restrict__ live_in_count) {
    // sample slot (all vertex data of a bounce is per slot): every slot, or the entries of the live list (then the dead slots are not touched at all: their
    // masks were cleared for the whole batch, nothing else of theirs is read again). The list's length is known on the device only, so the grid is a fixed
    // number of blocks that stride over it (no launch of 320 k blocks of which 14 k find work); without a list the loop runs once.
    const int n_items = live_in ? (int)*live_in_count : NV;
    for (int t0 = blockIdx.x * blockDim.x; t0 < n_items; t0 += gridDim.x * blockDim.x) {
    const int t_ = t0 + (int)threadIdx.x;
    const int pi = live_in ? (t_ < n_items ? live_in[t_] : NV) : t_;
    uint32_t mask = 0;  // bit0 NEE shadow ray, bit1 BSDF shadow ray, bit2 continuation ray
    v3 sp = V3(0.f), nee_dir = V3(0.f), bsdf_dir = V3(0.f), next_dir = V3(0.f);
    if (pi < NV) {
        const int k_ = pi / N, px = pi - k_ * N;
        v3 thr = V3(P.prd[5 * (size_t)pi], P.prd[5 * (size_t)pi + 1], P.prd[5 * (size_t)pi + 2]);
        float specularBounce = P.prd[5 * (size_t)pi + 3];
        float is_stop = P.prd[5 * (size_t)pi + 4];
        P.new_occ[pi] = 0.f; P.prd[5 * (size_t)pi + 4] = 1.f;
        if (bounce_count == 0) { thr = V3(1.0f); specularBounce = 0.f; is_stop = 0.f; P.prd[5 * (size_t)pi] = 1.f; P.prd[5 * (size_t)pi + 1] = 1.f; P.prd[5 * (size_t)pi + 2] = 1.f; P.prd[5 * (size_t)pi + 3] = 0.f; }
        v3 cv = V3(0.f), dcv = V3(0.f), scv = V3(0.f);
        if (!(is_stop > 0.f)) {
            Vertex v = load_vertex(P, pi);
            sp = v.pos;
            uint32_t sg = seed_generator((uint32_t)(px % fx), (uint32_t)(px / fx + y_off), slot_frame(frameIndex, k_, first_is_zero));
            if (P.occ[pi] > 0.1f) {
                shade::Lobes L = shade::lobes(v.diffuse, v.rough, v.metallic, v.rd, v.n);
                float lightPdf = 0.0f, scatteringPdf = 0.0f;
                float r0 = rnd(sg), r1 = rnd(sg);  // float2(sampleNext1D, sampleNext1D) in source order (:745)
                v3 samp_dir = V3(0.f), Li = V3(0.f); float samp_pdf = 0.f; v2 luv; bool samp_valid = false;
                {
                    v3 d; float p;
                    if (sample_li(E, r0, r1, d, p, luv)) { samp_valid = true; samp_dir = d; samp_pdf = p; Li = env_le(ngp_dir(d), E.tex, E.W, E.H) / p; }
                }
                shade::Frame fr = shade::create_frame(v.n);
                const v3 wi = shade::to_local(fr, -v.rd);
                lightPdf = samp_pdf;
                const v3 diffuse_col = v.diffuse * (1.0f - v.metallic);
                float* pd = pend + 18 * (size_t)pi;
                if (samp_valid && lightPdf > 0 && !is_black(Li)) {
                    v3 diff_f = V3(0.f), spec_f = V3(0.f), total_f = V3(0.f);
                    v3 wo = shade::to_local(fr, samp_dir);
                    if (!is_black(v.n)) {
                        if (L.pD > 0.f) diff_f = shade::diffuse_light(wi, wo);
                        if (L.pS > 0.f) spec_f = shade::specular_eval(wi, wo, L.specular, L.alpha);
                        total_f = diffuse_col * diff_f + spec_f;
                        diff_f = diffuse_col * diff_f;
                        scatteringPdf = shade::falcor_pdf(L.pD, L.pS, wi, wo, L.alpha);
                    }
                    if (!is_black(total_f)) {
                        nee_dir = normalize(samp_dir);
                        mask |= 1u;
                        float mis = mr_div(lightPdf * lightPdf, lightPdf * lightPdf + scatteringPdf * scatteringPdf);  // power_heuristic
                        v3 a = thr * total_f * Li * mis, b = thr * diff_f * Li * mis, c = thr * spec_f * Li * mis;
                        pd[0] = a.x; pd[1] = a.y; pd[2] = a.z; pd[3] = b.x; pd[4] = b.y; pd[5] = b.z; pd[6] = c.x; pd[7] = c.y; pd[8] = c.z;
                    }
                }
                uint32_t sampledSpecular = 0;
                if (!is_black(v.n)) {
                    v3 m_wi; float m_pdf; v3 dummy;
                    bool valid = shade::falcor_sample<false>(L.pD, L.pS, wi, m_wi, m_pdf, sampledSpecular, dummy, sg, L.alpha, L.specular, diffuse_col);
                    if (valid) {
                        v3 bdw = V3(1.0f), bsw = V3(1.0f);
                        if (L.pD > 0.f) bdw = shade::diffuse_light(wi, m_wi);
                        if (L.pS > 0.f) bsw = shade::specular_eval(wi, m_wi, L.specular, L.alpha);
                        v3 bw = diffuse_col * bdw + bsw;
                        m_wi = shade::to_global(fr, m_wi);
                        scatteringPdf = m_pdf;
                        v3 f = bw / m_pdf, diff_f = diffuse_col * bdw / m_pdf, spec_f = bsw / m_pdf;
                        f = f * m_pdf; diff_f = diff_f * m_pdf; spec_f = spec_f * m_pdf;
                        v3 safe_wi = normalize(m_wi);
                        if (!is_black(f) && scatteringPdf > 0) {
                            float weight = 1.0f; bool lightZero = false;
                            if (sampledSpecular == 0) {
                                lightPdf = pdf_li(E, safe_wi);
                                if (lightPdf == 0.0f) lightZero = true;
                                weight = mr_div(scatteringPdf * scatteringPdf, scatteringPdf * scatteringPdf + lightPdf * lightPdf);
                            }
                            bsdf_dir = safe_wi;
                            mask |= 2u;
                            v3 Le = env_le(ngp_dir(safe_wi), E.tex, E.W, E.H);  // used only if the ray escapes
                            v3 a = V3(0.f), b = V3(0.f), c = V3(0.f);
                            if (!is_black(Le) && !lightZero) {
                                const v3 Tr = V3(1.f);
                                a = thr * f * Le * Tr * weight / scatteringPdf;
                                b = thr * diff_f * Le * Tr * weight / scatteringPdf;
                                c = thr * spec_f * Le * Tr * weight / scatteringPdf;
                            }
                            pd[9] = a.x; pd[10] = a.y; pd[11] = a.z; pd[12] = b.x; pd[13] = b.y; pd[14] = b.z; pd[15] = c.x; pd[16] = c.y; pd[17] = c.z;
                        }
                    }
                }
                if (next_bounce_gen(P, pi, max_bounce, bounce_count, fr, L, wi, diffuse_col, thr, sg, next_dir)) mask |= 4u;
            } else {
                if (bounce_count == 0) cv = cv + thr * env_le(ngp_dir(v.rd), E.tex, E.W, E.H);
                else if (specularBounce > 0.f) { v3 e = thr * env_le(ngp_dir(v.rd), E.tex, E.W, E.H); cv = cv + e; scv = scv + e; }
                P.prd[5 * (size_t)pi + 4] = 1.f;
            }
        }
        // sparse (mirres_render's batches): most slots of a bounce are dead paths, so the three colours are written only where there is something
        // to say — an environment pick-up (bit 4 of the mask) — and k_bounce_resolve / k_pt_reduce read them only where bits 0, 1 or 4 are set
        const bool has_c = !is_black(cv) || !is_black(dcv) || !is_black(scv);
        if (has_c) mask |= 16u;
        if (!sparse || has_c) { st3(color, pi, cv); st3(diff_color, pi, dcv); st3(spec_color, pi, scv); }
    }
    const uint32_t na = (mask & 1u) + ((mask >> 1) & 1u);
    uint32_t base = block_append(qa_count, na > 0, na);
    uint32_t cs = block_append(qc_count, (mask & 4u) != 0);
    if (mask & 1u) put_ray(qa, base, sp, nee_dir, vis_near);
    if (mask & 2u) put_ray(qa, base + (mask & 1u), sp, bsdf_dir, vis_near);
    if (mask & 4u) put_ray(qc, cs, sp, next_dir, vis_near);
    if (pi < NV) { slot_a[pi] = na ? (int32_t)base : -1; mask_out[pi] = mask; slot_c[pi] = (mask & 4u) ? (int32_t)cs : -1; }
    }
}

template <bool ACC>
MR_DEV bool bounce_resolve_slot(const mirres_path_t& P, int pi, const int32_t* __restrict__ slot_a, const uint32_t* __restrict__ mask_in,
                                const int32_t* __restrict__ slot_c, const int32_t* __restrict__ hit, const HitRec* __restrict__ rec,
                                const float* __restrict__ pend, float* __restrict__ color, float* __restrict__ diff_color,
                                float* __restrict__ spec_color, float* __restrict__ acc_c, float* __restrict__ acc_d, float* __restrict__ acc_s, int sparse) {
    const uint32_t mask = mask_in[pi];
    v3 cv = V3(0.f), dcv = V3(0.f), scv = V3(0.f);
    if (!sparse || (mask & 16u)) { cv = ld3(color, pi); dcv = ld3(diff_color, pi); scv = ld3(spec_color, pi); }
    if (mask & 3u) {
        const float* pd = pend + 18 * (size_t)pi;
        int s = slot_a[pi];
        if (mask & 1u) {
            if (!hit[s]) { cv = cv + V3(pd[0], pd[1], pd[2]); dcv = dcv + V3(pd[3], pd[4], pd[5]); scv = scv + V3(pd[6], pd[7], pd[8]); }
            s++;
        }
        if (mask & 2u) {
            if (!hit[s]) { cv = cv + V3(pd[9], pd[10], pd[11]); dcv = dcv + V3(pd[12], pd[13], pd[14]); scv = scv + V3(pd[15], pd[16], pd[17]); }
        }
        st3(color, pi, cv); st3(diff_color, pi, dcv); st3(spec_color, pi, scv);
    }
    bool alive = false;
    if (mask & 4u) alive = next_bounce_resolve(P, pi, rec, slot_c[pi]);
    if (ACC) {  // renderer_restir.py:420-422 / 450-452 fused
        st3(acc_c, pi, ld3(acc_c, pi) + cv); st3(acc_d, pi, ld3(acc_d, pi) + dcv); st3(acc_s, pi, ld3(acc_s, pi) + scv);
    }
    return alive;
}
template <bool ACC>
__global__ void __launch_bounds__(MR_BLOCK) k_bounce_resolve(mirres_path_t P, int N, const int32_t* __restrict__ slot_a, const uint32_t* __restrict__ mask_in,
                                                             const int32_t* __restrict__ slot_c, const int32_t* __restrict__ hit, const HitRec* __restrict__ rec,
                                                             const float* __restrict__ pend, float* __restrict__ color, float* __restrict__ diff_color,
                                                             float* __restrict__ spec_color, float* __restrict__ acc_c, float* __restrict__ acc_d, float* __restrict__ acc_s, int sparse,
                                                             const int32_t* __restrict__ live_in, const uint32_t* __restrict__ live_in_count,
                                                             int32_t* __restrict__ live_out, uint32_t* __restrict__ live_out_count) {
    const int n_items = live_in ? (int)*live_in_count : N;
    for (int t0 = blockIdx.x * blockDim.x; t0 < n_items; t0 += gridDim.x * blockDim.x) {     // a fixed grid strides over the live list (see k_bounce_gen)
        const int t_ = t0 + (int)threadIdx.x;
        const int pi = live_in ? (t_ < n_items ? live_in[t_] : N) : t_;
        bool alive = false;
        if (pi < N) alive = bounce_resolve_slot<ACC>(P, pi, slot_a, mask_in, slot_c, hit, rec, pend, color, diff_color, spec_color, acc_c, acc_d, acc_s, sparse);
        if (live_out) {
            const uint32_t o = block_append(live_out_count, alive);
            if (alive) live_out[o] = pi;
        }
    }
}


static EnvD envh(const mirres_env_t* e) { EnvD E; E.tex = e->tex; E.W = e->Wc; E.H = e->Hc; E.pdf = e->pdf; E.cdf = e->cdf; E.mpdf = e->mpdf; E.mcdf = e->mcdf; return E; }

// internal launchers shared with render.hip
int launch_final_shading(const mirres_env_t* env, const float* occ, const float* normal, const float* ray_dir, const float* kd, const float* rm,
                         const float* fdir, const float* fdist, const float* fLi, int N, float* color, float* dl, float* sl, bool acc, hipStream_t s) {
    const int grd = grid_for(N, MR_BLOCK);
    if (acc) k_final_shading<true><<<grd, MR_BLOCK, 0, s>>>(envh(env), occ, normal, ray_dir, kd, rm, fdir, fdist, fLi, N, color, dl, sl);
    else k_final_shading<false><<<grd, MR_BLOCK, 0, s>>>(envh(env), occ, normal, ray_dir, kd, rm, fdir, fdist, fLi, N, color, dl, sl);
    MR_LAUNCH_CHECK("final_shading");
    return 0;
}
static PtQueues ctx_queues(mirres_ctx* ctx) {
    PtQueues q; q.any_rays = ctx->any_rays; q.any_hit = ctx->any_hit; q.cl_rays = ctx->cl_rays; q.cl_hit = ctx->cl_hit; q.counters = ctx->counters;
    q.slot_a = ctx->slot_a; q.mask_a = ctx->mask_a; q.slot_c = ctx->slot_c; q.pend = ctx->pend; q.N = (int)ctx->N; q.NV = (int)ctx->N; q.first_sample_is_zero = 0; q.lane = 0;
    q.live[0] = q.live[1] = nullptr; q.live_cur = 0;
    return q;
}
int launch_new_dir(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_path_t* p, uint32_t frameIndex, uint32_t bounce_count, hipStream_t s, const PtQueues* qq) {
    const PtQueues Q = qq ? *qq : ctx_queues(ctx);
    const int NV = Q.NV, grd = grid_for(NV, MR_BLOCK);
    MR_HIP(hipMemsetAsync(&Q.counters[1], 0, sizeof(uint32_t), s));
    k_new_dir_gen<<<grid_for(NV, MR_GEN_BLOCK), MR_GEN_BLOCK, 0, s>>>(*p, ctx->cfg.max_bounce, ctx->cfg.vis_near, frameIndex, bounce_count, ctx->fx, Q.N, NV, Q.first_sample_is_zero, ctx->y_off,
                                                                       Q.cl_rays, &Q.counters[1], Q.slot_c);
    int rc = trace_closest_q(ctx, bvh, Q.cl_rays, &Q.counters[1], (size_t)NV, Q.cl_hit, s, Q.lane);
    if (rc) return rc;
    if (Q.live[0]) MR_HIP(hipMemsetAsync(&Q.counters[3 + Q.live_cur], 0, sizeof(uint32_t), s));
    k_new_dir_resolve<<<grid_for(NV, MR_BLOCK * MR_RES_PER), MR_BLOCK, 0, s>>>(*p, NV, Q.slot_c, Q.cl_hit, Q.live[0] ? Q.live[Q.live_cur] : nullptr, &Q.counters[3 + Q.live_cur]);
    MR_LAUNCH_CHECK("pt_new_dir");
    return 0;
}
int launch_bounce(mirres_ctx* ctx, mirres_bvh* bvh, const mirres_env_t* env, const mirres_path_t* p, uint32_t frameIndex, uint32_t bounce_count, float* color,
                  float* dc, float* sc, float* acc_c, float* acc_d, float* acc_s, hipStream_t s, const PtQueues* qq) {
    const PtQueues Q = qq ? *qq : ctx_queues(ctx);
    const int NV = Q.NV, grd = grid_for(NV, MR_BLOCK);
    // live lists (batched frames): this vertex works on list live_cur and leaves the survivors in the other one
    const int32_t* lin = Q.live[0] ? Q.live[Q.live_cur] : nullptr; const uint32_t* lin_n = &Q.counters[3 + Q.live_cur];
    int32_t* lout = Q.live[0] ? Q.live[Q.live_cur ^ 1] : nullptr; uint32_t* lout_n = &Q.counters[3 + (Q.live_cur ^ 1)];
    MR_HIP(hipMemsetAsync(&Q.counters[0], 0, 2 * sizeof(uint32_t), s));
    // list mode: 16 blocks per CU stride over the device-side list; otherwise one thread per slot
    const int ggen = lin ? min(grid_for(NV, MR_BGEN_BLOCK), 256 * 16) : grid_for(NV, MR_BGEN_BLOCK), gres = lin ? min(grd, 256 * 16) : grd;
    k_bounce_gen<<<ggen, MR_BGEN_BLOCK, 0, s>>>(*p, envh(env), ctx->cfg.max_bounce, ctx->cfg.vis_near, frameIndex, bounce_count, ctx->fx, Q.N, NV, Q.first_sample_is_zero, ctx->y_off, qq ? 1 : 0,
                                                                      color, dc, sc, Q.any_rays, &Q.counters[0], Q.cl_rays, &Q.counters[1], Q.slot_a, Q.mask_a, Q.slot_c, Q.pend, lin, lin_n);
    int rc = trace_any_q(ctx, bvh, Q.any_rays, &Q.counters[0], 2 * (size_t)NV, Q.any_hit, s, Q.lane); if (rc) return rc;
    rc = trace_closest_q(ctx, bvh, Q.cl_rays, &Q.counters[1], (size_t)NV, Q.cl_hit, s, Q.lane); if (rc) return rc;
    if (lout) MR_HIP(hipMemsetAsync(lout_n, 0, sizeof(uint32_t), s));
    if (acc_c) k_bounce_resolve<true><<<gres, MR_BLOCK, 0, s>>>(*p, NV, Q.slot_a, Q.mask_a, Q.slot_c, Q.any_hit, Q.cl_hit, Q.pend, color, dc, sc, acc_c, acc_d, acc_s, qq ? 1 : 0, lin, lin_n, lout, lout_n);
    else k_bounce_resolve<false><<<gres, MR_BLOCK, 0, s>>>(*p, NV, Q.slot_a, Q.mask_a, Q.slot_c, Q.any_hit, Q.cl_hit, Q.pend, color, dc, sc, nullptr, nullptr, nullptr, qq ? 1 : 0, lin, lin_n, lout, lout_n);
    MR_LAUNCH_CHECK("pt_bounce");
    return 0;
}

}  // namespace mr

using namespace mr;

extern "C" {

int mirres_final_shading(mirres_ctx_t* ctx, const mirres_env_t* env, const float* occ, const float* normal, const float* ray_dir,
                         const float* kd, const float* rough_metal, const float* final_dir, const float* final_dist, const float* final_Li,
                         float* color, float* diff_light, float* spec_light, void* stream) {
    if (!ctx || !env || !occ || !normal || !ray_dir || !kd || !rough_metal || !final_dir || !final_dist || !final_Li || !color || !diff_light || !spec_light) {
        set_error("mirres_final_shading: null"); return MIRRES_E_ARG;
    }
    return launch_final_shading(env, occ, normal, ray_dir, kd, rough_metal, final_dir, final_dist, final_Li, (int)ctx->N, color, diff_light, spec_light, false,
                                (hipStream_t)stream);
}

int mirres_pt_new_dir(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_path_t* p, uint32_t frameIndex, uint32_t bounce_count, void* stream) {
    if (!ctx || !bvh || !p) { set_error("mirres_pt_new_dir: null"); return MIRRES_E_ARG; }
    return launch_new_dir(ctx, bvh, p, frameIndex, bounce_count, (hipStream_t)stream, nullptr);
}

int mirres_pt_bounce(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_env_t* env, const mirres_path_t* p, uint32_t frameIndex,
                     uint32_t bounce_count, float* color, float* diff_color, float* spec_color, void* stream) {
    if (!ctx || !bvh || !env || !p || !color || !diff_color || !spec_color) { set_error("mirres_pt_bounce: null"); return MIRRES_E_ARG; }
    return launch_bounce(ctx, bvh, env, p, frameIndex, bounce_count, color, diff_color, spec_color, nullptr, nullptr, nullptr, (hipStream_t)stream, nullptr);
}

}  // extern "C"
