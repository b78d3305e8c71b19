// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header). PARITY UNPINNED.
// The two BRDF libraries of the reference, kept separate on purpose:
//   rt::  = utils/brdf.slang    (reservoir target function / candidate pdfs; separable Smith, scalar Fresnel)
//   sh::  = utils/brdfDi.slang  (shading + path tracing; height-correlated Smith, RGB Fresnel)
#pragma once
#include "orc_math.hpp"

namespace orc {

// shared small pieces (identical text in both libraries)
static inline f3 perp_stark(f3 u) {  // brdf.slang:1-13
    f3 a = mk3(fabsf(u.x), fabsf(u.y), fabsf(u.z));
    uint32_t uyx = (a.x - a.y) < 0 ? 1 : 0;
    uint32_t uzx = (a.x - a.z) < 0 ? 1 : 0;
    uint32_t uzy = (a.y - a.z) < 0 ? 1 : 0;
    uint32_t xm = uyx & uzx;
    uint32_t ym = (1 ^ xm) & uzy;
    uint32_t zm = 1 ^ (xm | ym);
    return normalize(cross(u, mk3((float)xm, (float)ym, (float)zm)));
}
static inline f3 to_local(f3 w, f3 N) { f3 B = perp_stark(N); f3 T = cross(B, N); return mk3(dot(B, w), dot(T, w), dot(N, w)); }
static inline f3 to_global(f3 w, f3 N) { f3 B = perp_stark(N); f3 T = cross(B, N); return B * w.x + T * w.y + N * w.z; }

static inline float fresnel_schlick(float f0, float f90, float c) { return f0 + (f90 - f0) * mrf_pow5(fmaxf(1 - c, 0)); }
static inline f3 fresnel_schlick3(f3 f0, float f90, float c) {
    float p = mrf_pow5(fmaxf(1 - c, 0));
    return mk3(f0.x + (f90 - f0.x) * p, f0.y + (f90 - f0.y) * p, f0.z + (f90 - f0.z) * p);
}
static inline float lambda_ggx(float alphaSqr, float c) {  // brdf.slang:34-40
    if (c <= 0) return 0;
    float c2 = c * c;
    float tan2 = fmaxf(1 - c2, 0) / c2;
    return 0.5f * (-1 + sqrtf(1 + alphaSqr * tan2));
}
static inline float ndf_ggx(float alpha, float c) {  // brdf.slang:42-49
    const float M_PI_F = 3.141592653589793f;
    float a2 = alpha * alpha;
    float d = ((c * a2 - c) * c + 1);
    return a2 / (d * d * M_PI_F);
}
static inline float g_separable(float alpha, float ci, float co) {
    float a2 = alpha * alpha;
    return 1 / ((1 + lambda_ggx(a2, ci)) * (1 + lambda_ggx(a2, co)));
}
static inline float g_correlated(float alpha, float ci, float co) {
    float a2 = alpha * alpha;
    return 1 / (1 + lambda_ggx(a2, ci) + lambda_ggx(a2, co));
}
static inline float pdf_ggx_ndf(float alpha, float c) { return ndf_ggx(alpha, c) * c; }

static inline f2 sample_disk_concentric(f2 u) {  // brdf.slang:76-96
    const float M_PI_4_F = 0.785398163397448309616f, M_PI_2_F = 1.57079632679489661923f;
    u = mk2(2.f * u.x - 1.f, 2.f * u.y - 1.f);
    if (u.x == 0.f && u.y == 0.f) return u;
    float phi, r;
    if (fabsf(u.x) > fabsf(u.y)) { r = u.x; phi = (u.y / u.x) * M_PI_4_F; }
    else { r = u.y; phi = M_PI_2_F - (u.x / u.y) * M_PI_4_F; }
    return mk2(r * mrf_cos(phi), r * mrf_sin(phi));
}
static inline f3 sample_cosine_hemisphere_concentric(f2 u, float& pdf) {
    const float M_1_PI_F = 0.31830988f;
    f2 d = sample_disk_concentric(u);
    float z = sqrtf(fmaxf(0.f, 1.f - dot(d, d)));
    pdf = z * M_1_PI_F;
    return mk3(d.x, d.y, z);
}
static inline f3 sample_ggx_ndf(float alpha, f2 u, float& pdf) {  // brdf.slang:113-124
    const float M_PI_F = 3.141592653589793f;
    float a2 = alpha * alpha;
    float phi = u.y * (2 * M_PI_F);
    float tan2 = a2 * u.x / (1 - u.x);
    float c = 1 / sqrtf(1 + tan2);
    float r = sqrtf(fmaxf(1 - c * c, 0));
    pdf = pdf_ggx_ndf(alpha, c);
    return mk3(mrf_cos(phi) * r, mrf_sin(phi) * r, c);
}

namespace rt {  // utils/brdf.slang:155-212
static inline float eval_brdf(f3 L, f3 V, f3 N, float ggxAlpha, float wd, float ws) {
    const float M_1_PI_F = 0.31830988f;
    float wsum = wd + ws;
    float mix = wsum > 1e-7f ? (wd / wsum) : 1.f;
    float NdotV = saturate(dot(N, V)), NdotL = saturate(dot(N, L));
    f3 H = normalize(V + L);
    float NdotH = saturate(dot(N, H)), LdotH = saturate(dot(L, H));
    float D = ndf_ggx(ggxAlpha, NdotH);
    float G = g_separable(ggxAlpha, NdotV, NdotL);
    float F = ws < 1e-8f ? 0.f : fresnel_schlick(ws, 1.f, LdotH) / ws;
    float diffuse = NdotL * M_1_PI_F;
    float specular = fmaxf(0.f, D * G * F / (4.f * NdotV));
    return NdotL > 0.f ? lerpf(specular, diffuse, mix) : 0.f;
}
static inline float eval_pdf_brdf(f3 dir, f3 V, f3 N, float ggxAlpha, float wd, float ws) {  // specularOnly=false
    const float M_1_PI_F = 0.31830988f;
    float wsum = wd + ws;
    float mix = wsum > 1e-7f ? (wd / wsum) : 1.f;
    float c = saturate(dot(N, dir));
    float diffusePdf = c * M_1_PI_F;
    f3 h = normalize(to_local(dir + V, N));
    float specularPdf = pdf_ggx_ndf(ggxAlpha, h.z) / (4.f * saturate(dot(h, to_local(V, N))));
    return c > 0.f ? lerpf(specularPdf, diffusePdf, mix) : 0.f;
}
static inline bool sample_brdf(f3 xi, f3& dir, f3 V, f3 N, float ggxAlpha, float wd, float ws) {
    float wsum = wd + ws;
    float mix = wsum > 1e-7f ? (wd / wsum) : 1.f;
    dir = mk3(0.f);
    float pdf;
    if (xi.x < mix) dir = to_global(sample_cosine_hemisphere_concentric(mk2(xi.y, xi.z), pdf), N);
    else { f3 h = sample_ggx_ndf(ggxAlpha, mk2(xi.y, xi.z), pdf); dir = reflect(-V, to_global(h, N)); }
    return dot(N, dir) > 0.f;
}
// res.slang:70-77
static inline float target(f3 emission, f3 ldir, f3 normal, f3 ray_dir, f3 brdf) {
    float w = eval_brdf(ldir, -ray_dir, normal, brdf.z, brdf.x, brdf.y);
    return fmaxf(0.f, luminance(emission) * w);
}
}  // namespace rt

namespace sh {  // utils/brdfDi.slang + helperDi.slang frame
struct Frame { f3 x, y, z; };
static inline Frame create_frame(f3 n) {  // helperDi.slang:9-28
    Frame f; f.z = n;
    float sign = (n.z > 0) ? 1.0f : -1.0f;
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    f.x = mk3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    f.y = mk3(b, sign + n.y * n.y * a, -n.y);
    return f;
}
static inline f3 to_local(const Frame& f, f3 v) { return mk3(dot(f.x, v), dot(f.y, v), dot(f.z, v)); }
static inline f3 to_global(const Frame& f, f3 v) { return f.x * v.x + f.y * v.y + f.z * v.z; }

static inline f3 diffuse_light(f3 wo, f3 wi) {  // brdfDi.slang:169-177
    const float M_1_PI_F = 0.31830988f;
    if (fminf(wo.z, wi.z) < 1e-6f) return mk3(0.f);
    return mk3(fmaxf(M_1_PI_F * wi.z, 0.0f));
}
static inline f3 diffuse_eval(f3 wo, f3 wi, f3 albedo) {  // :138-146
    const float M_1_PI_F = 0.31830988f;
    if (fminf(wo.z, wi.z) < 1e-6f) return mk3(0.f);
    return M_1_PI_F * albedo * wi.z;
}
static inline float diffuse_eval_pdf(f3 wo, f3 wi) {  // :148-155
    const float M_1_PI_F = 0.31830988f;
    if (fminf(wo.z, wi.z) < 1e-6f) return 0.f;
    return M_1_PI_F * wi.z;
}
// SpecularReflection_eval(activeLobes=true, allowDeltaEval=false)  :179-199
static inline f3 specular_eval(f3 wo, f3 wi, f3 albedo, float alpha) {
    if (fminf(wo.z, wi.z) < 1e-6f) return mk3(0.f);
    if (alpha == 0.f) return mk3(0.f);
    f3 h = normalize(wo + wi);
    float woDotH = dot(wo, h);
    float D = ndf_ggx(alpha, h.z);
    float G = g_correlated(alpha, wo.z, wi.z);
    f3 F = fresnel_schlick3(albedo, 1, woDotH);
    return F * D * G * 0.25f / wo.z;
}
// SpecularReflection_evalPdf(activeLobes=true, allowDeltaEval=false)  :201-221
static inline float specular_eval_pdf(f3 wo, f3 wi, float alpha) {
    if (fminf(wo.z, wi.z) < 1e-6f) return 0.f;
    if (alpha == 0.f) return 0.f;
    f3 h = normalize(wo + wi);
    float woDotH = dot(wo, h);
    return pdf_ggx_ndf(alpha, h.z) / (4.f * woDotH);
}
// DiffuseReflection_sample :157-171  (1 burn + 2 draws)
static inline bool diffuse_sample(f3 wo, f3& wi, float& pdf, uint32_t& sg) {
    next1d(sg);
    float a = next1d(sg), b = next1d(sg);
    wi = sample_cosine_hemisphere_concentric(mk2(a, b), pdf);
    return !(fminf(wo.z, wi.z) < 1e-6f);
}
// SpecularReflection_sample(activeLobes=true, allowDeltaEval=false) :223-259
static inline bool specular_sample(float alpha, f3 wo, f3& wi, float& pdf, uint32_t& sg) {
    wi = mk3(0.f); pdf = 0.f;
    if (wo.z < 1e-6f) return false;
    next1d(sg);
    if (alpha == 0.f) return false;
    float a = next1d(sg), b = next1d(sg);
    f3 h = sample_ggx_ndf(alpha, mk2(a, b), pdf);
    float woDotH = dot(wo, h);
    wi = 2.f * woDotH * h - wo;
    if (wi.z < 1e-6f) return false;
    pdf = specular_eval_pdf(wo, wi, alpha);
    return true;
}
// FalcorBRDF_eval(activeLobes=true, allowDeltaEval=false) :261-270
static inline f3 falcor_eval(float pD, float pS, float alpha, f3 spec_albedo, f3 diff_albedo, f3 wo, f3 wi) {
    f3 r = mk3(0.f);
    if (pD > 0.f) r += diffuse_eval(wo, wi, diff_albedo);
    if (pS > 0.f) r += specular_eval(wo, wi, spec_albedo, alpha);
    return r;
}
// FalcorBRDF_evalPdf(activeLobes=true, allowDeltaEval=false) :272-282
static inline float falcor_eval_pdf(float pD, float pS, f3 wo, f3 wi, float alpha) {
    float pdf = 0.f;
    if (pD > 0.f) pdf += pD * diffuse_eval_pdf(wo, wi);
    if (pS > 0.f) pdf += pS * specular_eval_pdf(wo, wi, alpha);
    return pdf;
}
// FalcorBRDF_sample (:285-328) when with_weight, FalcorBRDF_sample_no_weight (:396-457) otherwise.
static inline bool falcor_sample(float pD, float pS, f3 wo, f3& wi, float& pdf, uint32_t& specularBounce, f3& weight,
                                 uint32_t& sg, float alpha, f3 spec_albedo, f3 diff_albedo, bool with_weight) {
    wi = mk3(0.f); weight = mk3(0.f); pdf = 0.f; specularBounce = 0;
    bool valid = false;
    float uSelect = next1d(sg);
    if (uSelect < pD) {
        valid = diffuse_sample(wo, wi, pdf, sg);
        if (with_weight) weight = falcor_eval(pD, pS, alpha, spec_albedo, diff_albedo, wo, wi);
        pdf *= pD;
        if (pS > 0.f) pdf += pS * specular_eval_pdf(wo, wi, alpha);
        if (with_weight) weight = weight / pdf;
    } else if (uSelect < pD + pS) {
        valid = specular_sample(alpha, wo, wi, pdf, sg);
        if (with_weight) weight = falcor_eval(pD, pS, alpha, spec_albedo, diff_albedo, wo, wi);
        pdf *= pS;
        float test_roughness = sqrtf(alpha);
        if (test_roughness > 0.15f) { if (pD > 0.f) pdf += pD * diffuse_eval_pdf(wo, wi); }
        else specularBounce = 1;
        if (with_weight) weight = weight / pdf;
    }
    return valid;
}

// lobe probabilities + alpha: FinalShading.slang:58-78 (same block at :184-203 and :723-743)
struct Lobes { float pD, pS, alpha; f3 specular; };
static inline Lobes lobes(f3 diffuse, float linearRoughness, float metallic, f3 ray_dir, f3 normal) {
    Lobes L;
    const float F0 = 0.04f;
    L.specular = mk3(F0) * (1.0f - metallic) + diffuse * metallic;
    float kMin = 0.01f * 0.01f;
    L.alpha = linearRoughness * linearRoughness;
    if (L.alpha < kMin) L.alpha = 0.f;
    float diffuseWeight = luminance(diffuse);
    float dielectric = (1.f - metallic) * (1.f - 0.f);
    L.pD = diffuseWeight * dielectric * (1.f - 0.f);
    float specularWeight = luminance(fresnel_schlick3(L.specular, 1.f, dot(-ray_dir, normal)));
    L.pS = specularWeight * (metallic + dielectric);
    float nf = L.pD + L.pS;
    if (nf > 0.f) { nf = 1.f / nf; L.pD *= nf; L.pS *= nf; }
    return L;
}
}  // namespace sh

}  // namespace orc
