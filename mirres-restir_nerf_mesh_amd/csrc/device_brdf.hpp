// device_brdf.hpp — the two BRDF models of the path, kept apart on purpose:
//   rtarget:: reservoir target function + candidate pdfs (utils/brdf.slang: separable Smith G, scalar Fresnel)
//   shade::   final shading + path tracing (utils/brdfDi.slang: height-correlated Smith G, RGB Fresnel)
#pragma once
#include "device_math.hpp"

namespace mr {

MR_DEV v3 perp_stark(v3 u) {  // brdf.slang:1-13
    float ax = fabsf(u.x), ay = fabsf(u.y), az = fabsf(u.z);
    uint32_t uyx = (ax - ay) < 0 ? 1 : 0, uzx = (ax - az) < 0 ? 1 : 0, uzy = (ay - az) < 0 ? 1 : 0;
    uint32_t xm = uyx & uzx, ym = (1 ^ xm) & uzy, zm = 1 ^ (xm | ym);
    return normalize(cross(u, V3((float)xm, (float)ym, (float)zm)));
}
struct Basis { v3 B, T, N; };
MR_DEV Basis basis(v3 N) { Basis b; b.B = perp_stark(N); b.T = cross(b.B, N); b.N = N; return b; }
MR_DEV v3 to_local(const Basis& b, v3 w) { return V3(dot(b.B, w), dot(b.T, w), dot(b.N, w)); }
MR_DEV v3 to_global(const Basis& b, v3 w) { return b.B * w.x + b.T * w.y + b.N * w.z; }

// evalFresnelSchlick's pow(max(1-c,0),5): mrf_pow5 (include/mirres_fmath.h: six operations, <= 2 ulp, the same bits on the host) instead of a library powf
// (~40 VALU instructions; 33 evaluations per pixel in the initial pass).
MR_DEV float pow5(float c) { return mrf_pow5(fmaxf(1 - c, 0)); }
MR_DEV float schlick(float f0, float f90, float c) { return f0 + (f90 - f0) * pow5(c); }
MR_DEV v3 schlick3(v3 f0, float f90, float c) { float p = pow5(c); return V3(f0.x + (f90 - f0.x) * p, f0.y + (f90 - f0.y) * p, f0.z + (f90 - f0.z) * p); }
MR_DEV float lambda_ggx(float a2, float c) {  // brdf.slang:34-40
    if (c <= 0) return 0;
    float c2 = c * c;
    float tan2 = mr_div(fmaxf(1 - c2, 0), c2);
    return 0.5f * (-1 + mr_sqrt(1 + a2 * tan2));
}
MR_DEV float ndf_ggx(float alpha, float c) {  // brdf.slang:42-49
    float a2 = alpha * alpha;
    float d = ((c * a2 - c) * c + 1);
    return mr_div(a2, d * d * 3.141592653589793f);
}
MR_DEV float g_separable(float alpha, float ci, float co) { float a2 = alpha * alpha; return mr_rcp((1 + lambda_ggx(a2, ci)) * (1 + lambda_ggx(a2, co))); }
MR_DEV float g_correlated(float alpha, float ci, float co) { float a2 = alpha * alpha; return mr_rcp(1 + lambda_ggx(a2, ci) + lambda_ggx(a2, co)); }
MR_DEV float pdf_ggx_ndf(float alpha, float c) { return ndf_ggx(alpha, c) * c; }

MR_DEV v2 disk_concentric(float ux, float uy) {  // brdf.slang:76-96
    ux = 2.f * ux - 1.f; uy = 2.f * uy - 1.f;
    if (ux == 0.f && uy == 0.f) return V2(ux, uy);
    float phi, r;
    if (fabsf(ux) > fabsf(uy)) { r = ux; phi = mr_div(uy, ux) * 0.785398163397448309616f; }
    else { r = uy; phi = 1.57079632679489661923f - mr_div(ux, uy) * 0.785398163397448309616f; }
    float sp, cp; mrf_sincos(phi, &sp, &cp);
    return V2(r * cp, r * sp);
}
MR_DEV v3 cosine_hemisphere(float ux, float uy, float& pdf) {
    v2 d = disk_concentric(ux, uy);
    float z = mr_sqrt(fmaxf(0.f, 1.f - dot(d, d)));
    pdf = z * 0.31830988f;
    return V3(d.x, d.y, z);
}
MR_DEV v3 sample_ggx_ndf(float alpha, float ux, float uy, float& pdf) {  // brdf.slang:113-124
    float a2 = alpha * alpha;
    float phi = uy * (2 * 3.141592653589793f);
    float tan2 = mr_div(a2 * ux, 1 - ux);
    float c = mr_rcp(mr_sqrt(1 + tan2));
    float r = mr_sqrt(fmaxf(1 - c * c, 0));
    pdf = pdf_ggx_ndf(alpha, c);
    float sp, cp; mrf_sincos(phi, &sp, &cp);
    return V3(cp * r, sp * r, c);
}

namespace rtarget {  // utils/brdf.slang:155-212, res.slang:70-91
// pixel context hoisted once per pixel: V = -ray_dir, the perp_stark basis of N, V in that basis
struct Ctx { v3 N, V; float alpha, wd, ws, mix; Basis b; v3 Vl; };
MR_DEV Ctx make_ctx(v3 N, v3 ray_dir, v3 brdf) {
    Ctx c; c.N = N; c.V = -ray_dir; c.alpha = brdf.z; c.wd = brdf.x; c.ws = brdf.y;
    float wsum = c.wd + c.ws;
    c.mix = wsum > 1e-7f ? mr_div(c.wd, wsum) : 1.f;
    c.b = basis(N); c.Vl = to_local(c.b, c.V);
    return c;
}
MR_DEV float eval_brdf(const Ctx& c, v3 L) {
    float NdotV = saturate(dot(c.N, c.V)), NdotL = saturate(dot(c.N, L));
    // No specular weight (metallic = 0: the reference's default --me_max 0, main.py:110, makes brdf_map.y = 0 in every pixel): F is the constant 0, so
    // `specular` = max(0, (D G 0) / (4 NdotV)) is +0 whatever D and G are (they are finite or NaN, never infinite: D = a2 / (d d pi) with d > 0 for a2 > 0, and a
    // NaN — a2 = 0 at NdotH = 1, or 0 / 0 at NdotV = 0 — is dropped by fmaxf), and lerp(+0, diffuse, mix) = (diffuse - 0) mix exactly. The six divisions and two
    // square roots of D, G, F and the quotient are skipped; the result has the same bits (the CPU checker takes the long way; full-size frames compare bit for bit).
    if (c.ws < 1e-8f) return NdotL > 0.f ? lerpf(0.f, NdotL * 0.31830988f, c.mix) : 0.f;
    v3 H = normalize(c.V + L);
    float NdotH = saturate(dot(c.N, H)), LdotH = saturate(dot(L, H));
    float D = ndf_ggx(c.alpha, NdotH);
    float G = g_separable(c.alpha, NdotV, NdotL);
    float F = c.ws < 1e-8f ? 0.f : mr_div(schlick(c.ws, 1.f, LdotH), c.ws);
    float diffuse = NdotL * 0.31830988f;
    float specular = fmaxf(0.f, mr_div(D * G * F, 4.f * NdotV));
    return NdotL > 0.f ? lerpf(specular, diffuse, c.mix) : 0.f;
}
MR_DEV float target(const Ctx& c, v3 emission, v3 L) { return fmaxf(0.f, luminance(emission) * eval_brdf(c, L)); }
MR_DEV float pdf_brdf(const Ctx& c, v3 dir) {
    float ct = saturate(dot(c.N, dir));
    float diffusePdf = ct * 0.31830988f;
    v3 h = normalize(to_local(c.b, dir + c.V));
    float specularPdf = mr_div(pdf_ggx_ndf(c.alpha, h.z), 4.f * saturate(dot(h, c.Vl)));
    return ct > 0.f ? lerpf(specularPdf, diffusePdf, c.mix) : 0.f;
}
MR_DEV bool sample_brdf(const Ctx& c, float xa, float xb, float xc, v3& dir) {
    float pdf;
    if (xa < c.mix) dir = to_global(c.b, cosine_hemisphere(xb, xc, pdf));
    else { v3 h = sample_ggx_ndf(c.alpha, xb, xc, pdf); dir = reflect(-c.V, to_global(c.b, h)); }
    return dot(c.N, dir) > 0.f;
}
}  // namespace rtarget

namespace shade {
struct Frame { v3 x, y, z; };
MR_DEV Frame create_frame(v3 n) {  // helperDi.slang:9-28
    Frame f; f.z = n;
    float sign = (n.z > 0) ? 1.0f : -1.0f;
    const float a = mr_div(-1.0f, sign + n.z);
    const float b = n.x * n.y * a;
    f.x = V3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    f.y = V3(b, sign + n.y * n.y * a, -n.y);
    return f;
}
MR_DEV v3 to_local(const Frame& f, v3 v) { return V3(dot(f.x, v), dot(f.y, v), dot(f.z, v)); }
MR_DEV v3 to_global(const Frame& f, v3 v) { return f.x * v.x + f.y * v.y + f.z * v.z; }

MR_DEV v3 diffuse_light(v3 wo, v3 wi) { if (fminf(wo.z, wi.z) < 1e-6f) return V3(0.f); return V3(fmaxf(0.31830988f * wi.z, 0.0f)); }  // brdfDi.slang:169-177
MR_DEV v3 diffuse_eval(v3 wo, v3 wi, v3 albedo) { if (fminf(wo.z, wi.z) < 1e-6f) return V3(0.f); return 0.31830988f * albedo * wi.z; }
MR_DEV float diffuse_pdf(v3 wo, v3 wi) { if (fminf(wo.z, wi.z) < 1e-6f) return 0.f; return 0.31830988f * wi.z; }
MR_DEV v3 specular_eval(v3 wo, v3 wi, v3 albedo, float alpha) {  // SpecularReflection_eval(activeLobes=true) :179-199
    if (fminf(wo.z, wi.z) < 1e-6f) return V3(0.f);
    if (alpha == 0.f) return V3(0.f);
    v3 h = normalize(wo + wi);
    float woDotH = dot(wo, h);
    float D = ndf_ggx(alpha, h.z);
    float G = g_correlated(alpha, wo.z, wi.z);
    v3 F = schlick3(albedo, 1, woDotH);
    return F * D * G * 0.25f / wo.z;
}
MR_DEV float specular_pdf(v3 wo, v3 wi, float alpha) {  // :201-221
    if (fminf(wo.z, wi.z) < 1e-6f) return 0.f;
    if (alpha == 0.f) return 0.f;
    v3 h = normalize(wo + wi);
    return mr_div(pdf_ggx_ndf(alpha, h.z), 4.f * dot(wo, h));
}
MR_DEV bool diffuse_sample(v3 wo, v3& wi, float& pdf, uint32_t& sg) {  // :157-171 (1 burn + 2 draws)
    rnd(sg);
    float a = rnd(sg), b = rnd(sg);
    wi = cosine_hemisphere(a, b, pdf);
    return !(fminf(wo.z, wi.z) < 1e-6f);
}
MR_DEV bool specular_sample(float alpha, v3 wo, v3& wi, float& pdf, uint32_t& sg) {  // :223-259
    wi = V3(0.f); pdf = 0.f;
    if (wo.z < 1e-6f) return false;
    rnd(sg);
    if (alpha == 0.f) return false;
    float a = rnd(sg), b = rnd(sg);
    v3 h = sample_ggx_ndf(alpha, a, b, pdf);
    float woDotH = dot(wo, h);
    wi = 2.f * woDotH * h - wo;
    if (wi.z < 1e-6f) return false;
    pdf = specular_pdf(wo, wi, alpha);
    return true;
}
MR_DEV v3 falcor_eval(float pD, float pS, float alpha, v3 spec_albedo, v3 diff_albedo, v3 wo, v3 wi) {  // :261-270
    v3 r = V3(0.f);
    if (pD > 0.f) r = r + diffuse_eval(wo, wi, diff_albedo);
    if (pS > 0.f) r = r + specular_eval(wo, wi, spec_albedo, alpha);
    return r;
}
MR_DEV float falcor_pdf(float pD, float pS, v3 wo, v3 wi, float alpha) {  // :272-282
    float pdf = 0.f;
    if (pD > 0.f) pdf += pD * diffuse_pdf(wo, wi);
    if (pS > 0.f) pdf += pS * specular_pdf(wo, wi, alpha);
    return pdf;
}
// FalcorBRDF_sample (:285-328, WEIGHT=true) / FalcorBRDF_sample_no_weight (:396-457, WEIGHT=false)
template <bool WEIGHT>
MR_DEV bool falcor_sample(float pD, float pS, v3 wo, v3& wi, float& pdf, uint32_t& specularBounce, v3& weight, uint32_t& sg, float alpha,
                          v3 spec_albedo, v3 diff_albedo) {
    wi = V3(0.f); weight = V3(0.f); pdf = 0.f; specularBounce = 0;
    bool valid = false;
    float uSelect = rnd(sg);
    if (uSelect < pD) {
        valid = diffuse_sample(wo, wi, pdf, sg);
        if (WEIGHT) weight = falcor_eval(pD, pS, alpha, spec_albedo, diff_albedo, wo, wi);
        pdf *= pD;
        if (pS > 0.f) pdf += pS * specular_pdf(wo, wi, alpha);
        if (WEIGHT) weight = weight / pdf;
    } else if (uSelect < pD + pS) {
        valid = specular_sample(alpha, wo, wi, pdf, sg);
        if (WEIGHT) weight = falcor_eval(pD, pS, alpha, spec_albedo, diff_albedo, wo, wi);
        pdf *= pS;
        if (mr_sqrt(alpha) > 0.15f) { if (pD > 0.f) pdf += pD * diffuse_pdf(wo, wi); }
        else specularBounce = 1;
        if (WEIGHT) weight = weight / pdf;
    }
    return valid;
}
// lobe probabilities (FinalShading.slang:58-78, repeated at :184-203 and :723-743)
struct Lobes { float pD, pS, alpha; v3 specular; };
MR_DEV Lobes lobes(v3 diffuse, float roughness, float metallic, v3 ray_dir, v3 normal) {
    Lobes L;
    L.specular = V3(0.04f) * (1.0f - metallic) + diffuse * metallic;
    L.alpha = roughness * roughness;
    if (L.alpha < 0.01f * 0.01f) L.alpha = 0.f;
    float dielectric = (1.f - metallic) * (1.f - 0.f);
    L.pD = luminance(diffuse) * dielectric * (1.f - 0.f);
    float sw = luminance(schlick3(L.specular, 1.f, dot(-ray_dir, normal)));
    L.pS = sw * (metallic + dielectric);
    float nf = L.pD + L.pS;
    if (nf > 0.f) { nf = mr_rcp(nf); L.pD *= nf; L.pS *= nf; }
    return L;
}
}  // namespace shade

}  // namespace mr
