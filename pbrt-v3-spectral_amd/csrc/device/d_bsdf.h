// d_bsdf.h -- device BSDF over the compiled lobe list (mi_material in HBM).
//
// A BSDF value is a 31-bin spectrum; to keep VGPR pressure flat the kernels never
// hold one. Each lobe evaluation is reduced to a few scalars (LobeEval: how to get
// f_lobe[bin] from the lobe's reflectance R[bin]) in a scalar phase, and the
// spectral phase streams bins: f[bin] = sum over lobes. The per-bin operation
// chains reproduce the reference's operator order so values match its rounding:
//   BSDF::f / Sample_f / Pdf      src/core/reflection.cpp:670-785
//   BxDFs, Fresnel                src/core/reflection.cpp:47-511, reflection.h:50-127
//   Trowbridge-Reitz              src/core/microfacet.cpp:165-184,238-344
//   Disney lobes                  src/materials/disney.cpp:61-356
#pragma once
#include "d_scene.h"

namespace dpt {

DEV float CosTheta(const V3 &w) { return w.z; }
DEV float Cos2Theta(const V3 &w) { return w.z * w.z; }
DEV float AbsCosTheta(const V3 &w) { return absf(w.z); }
DEV float Sin2Theta(const V3 &w) { return maxf(0.f, 1.f - Cos2Theta(w)); }
DEV float SinTheta(const V3 &w) { return __builtin_sqrtf(Sin2Theta(w)); }
DEV float TanTheta(const V3 &w) { return SinTheta(w) / CosTheta(w); }
DEV float Tan2Theta(const V3 &w) { return Sin2Theta(w) / Cos2Theta(w); }
DEV float CosPhi(const V3 &w) { float s = SinTheta(w); return (s == 0) ? 1 : clampf(w.x / s, -1, 1); }
DEV float SinPhi(const V3 &w) { float s = SinTheta(w); return (s == 0) ? 0 : clampf(w.y / s, -1, 1); }
DEV float Cos2Phi(const V3 &w) { return CosPhi(w) * CosPhi(w); }
DEV float Sin2Phi(const V3 &w) { return SinPhi(w) * SinPhi(w); }
DEV V3 Reflect(const V3 &wo, const V3 &n) { return -wo + 2 * Dot(wo, n) * n; }
DEV bool Refract(const V3 &wi, const V3 &n, float eta, V3 *wt) {
    float cosThetaI = Dot(n, wi);
    float sin2ThetaI = maxf(0.f, 1 - cosThetaI * cosThetaI);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = __builtin_sqrtf(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}
DEV bool SameHemisphere(const V3 &w, const V3 &wp) { return w.z * wp.z > 0; }

DEV float FrDielectric(float cosThetaI, float etaI, float etaT) {  // reflection.cpp:47-69
    cosThetaI = clampf(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) { float t = etaI; etaI = etaT; etaT = t; cosThetaI = absf(cosThetaI); }
    float sinThetaI = __builtin_sqrtf(maxf(0.f, 1 - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    float cosThetaT = __builtin_sqrtf(maxf(0.f, 1 - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}

DEV void ConcentricSampleDisk(float u0, float u1, float *dx, float *dy) {  // sampling.cpp:113-130
    float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    if (ox == 0 && oy == 0) { *dx = 0; *dy = 0; return; }
    float theta, r;
    if (absf(ox) > absf(oy)) { r = ox; theta = kPiOver4 * (oy / ox); }
    else { r = oy; theta = kPiOver2 - kPiOver4 * (ox / oy); }
    *dx = r * cosF(theta);
    *dy = r * sinF(theta);
}
DEV V3 CosineSampleHemisphere(float u0, float u1) {
    float dx, dy;
    ConcentricSampleDisk(u0, u1, &dx, &dy);
    float z = __builtin_sqrtf(maxf(0.f, 1 - dx * dx - dy * dy));
    return V3(dx, dy, z);
}

// ---- Trowbridge-Reitz (GGX), visible-normal sampling
struct TRDist {
    float alphax, alphay;
    bool separableG;
    DEV float D(const V3 &wh) const {
        float tan2Theta = Tan2Theta(wh);
        if (isinff(tan2Theta)) return 0.;
        const float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
        float e = (Cos2Phi(wh) / (alphax * alphax) + Sin2Phi(wh) / (alphay * alphay)) * tan2Theta;
        return 1 / (kPi * alphax * alphay * cos4Theta * (1 + e) * (1 + e));
    }
    DEV float Lambda(const V3 &w) const {
        float absTanTheta = absf(TanTheta(w));
        if (isinff(absTanTheta)) return 0.;
        float alpha = __builtin_sqrtf(Cos2Phi(w) * alphax * alphax + Sin2Phi(w) * alphay * alphay);
        float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
        return (-1 + __builtin_sqrtf(1.f + alpha2Tan2Theta)) / 2;
    }
    DEV float G1(const V3 &w) const { return 1 / (1 + Lambda(w)); }
    DEV float G(const V3 &wo, const V3 &wi) const {
        if (separableG) return G1(wo) * G1(wi);
        return 1 / (1 + Lambda(wo) + Lambda(wi));
    }
    DEV float Pdf(const V3 &wo, const V3 &wh) const { return D(wh) * G1(wo) * AbsDot(wo, wh) / AbsCosTheta(wo); }
    DEV static void Sample11(float cosTheta, float U1, float U2, float *slope_x, float *slope_y) {
        if ((double)cosTheta > .9999) {
            float r = (float)__builtin_sqrt((double)(U1 / (1 - U1)));
            float phi = (float)(6.28318530718 * (double)U2);
            *slope_x = r * (float)cos((double)phi);
            *slope_y = r * (float)sin((double)phi);
            return;
        }
        float sinTheta = __builtin_sqrtf(maxf(0.f, 1.f - cosTheta * cosTheta));
        float tanTheta = sinTheta / cosTheta;
        float a = 1 / tanTheta;
        float G1 = 2 / (1 + __builtin_sqrtf(1.f + 1.f / (a * a)));
        float A = 2 * U1 / G1 - 1;
        float tmp = 1.f / (A * A - 1.f);
        if ((double)tmp > 1e10) tmp = (float)1e10;
        float B = tanTheta;
        float D = __builtin_sqrtf(maxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
        float slope_x_1 = B * tmp - D;
        float slope_x_2 = B * tmp + D;
        *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
        float S;
        if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
        else { S = -1.f; U2 = 2.f * (.5f - U2); }
        float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        *slope_y = S * z * __builtin_sqrtf(1.f + *slope_x * *slope_x);
    }
    DEV V3 Sample_wh(const V3 &wo, float u0, float u1) const {
        bool flip = wo.z < 0;
        V3 wi = flip ? -wo : wo;
        V3 wiStretched = Normalize(V3(alphax * wi.x, alphay * wi.y, wi.z));
        float slope_x, slope_y;
        Sample11(CosTheta(wiStretched), u0, u1, &slope_x, &slope_y);
        float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
        slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = alphax * slope_x;
        slope_y = alphay * slope_y;
        V3 wh = Normalize(V3(-slope_x, -slope_y, 1.f));
        if (flip) wh = -wh;
        return wh;
    }
};

DEV float SchlickWeight(float cosTheta) { float m = clampf(1 - cosTheta, 0, 1); return (m * m) * (m * m) * m; }
DEV float FrSchlickF(float R0, float cosTheta) { return lerpf(SchlickWeight(cosTheta), R0, 1); }
DEV float GTR1(float cosTheta, float alpha) {
    float alpha2 = alpha * alpha;
    return (alpha2 - 1) / (kPi * logF(alpha2) * (1 + (alpha2 - 1) * cosTheta * cosTheta));
}
DEV float smithG_GGX(float cosTheta, float alpha) {
    float alpha2 = alpha * alpha;
    float cosTheta2 = cosTheta * cosTheta;
    return 1 / (cosTheta + (float)__builtin_sqrt((double)(alpha2 + cosTheta2 - alpha2 * cosTheta2)));
}

// ---- scalar description of one lobe's f for a fixed (wo, wi)
enum LobeKind : int {
    LK_NONE = 0,
    LK_MUL1,      // R*a
    LK_MUL2,      // (R*a)*b
    LK_MUL3,      // ((R*a)*b)*c
    LK_MUL3_DIV,  // (((R*a)*b)*c)/d
    LK_MUL1_DIV,  // (R*a)/d
    LK_MUL2_DIV,  // ((R*a)*b)/d
    LK_MICRO_DISNEY,  // (((R*a)*b)*F[bin])/d, F = lerp(c, FrDielectric=e, FrSchlick(S[bin], w=f))
    LK_MTRANS,    // ((1-a)*R)*b
    LK_CONST,     // a
    LK_MICRO_CONDUCTOR,  // (((R*a)*b)*F[bin])/d, F = FrConductor(cos, 1, S[bin], K[bin]); c=cos^2, e=sin^2, f=2cos
    LK_FBLEND     // FresnelBlend: (((R*f)*(1-S))*a)*b + (S + (1-S)*c)*e
};
struct LobeEval {
    int kind;  // LobeKind | LK_FASTDIV when DivBy's fast form applies to d
    int lobe;  // index into material.bxdf
    float a, b, c, d, e, f;
    float r;   // 1/d for the *_DIV kinds
};
#define LK_FASTDIV 0x100
DEV void SetDivisor(LobeEval &le, float d) {
    Divisor v = MakeDivisor(d);
    le.d = d;
    le.r = v.r;
    if (v.fast) le.kind |= LK_FASTDIV;
}
DEV float LobeDiv(float x, const LobeEval &le) {
    Divisor v;
    v.d = le.d; v.r = le.r; v.fast = (le.kind & LK_FASTDIV) != 0;
    return DivBy(x, v);
}

// One bin of FrConductor(cosThetaI, Spectrum(1), etaT, k) (reflection.cpp:71-94) in the reference's
// operator order; cos2 = cos^2, sin2 = (Float)(1. - cos2), twoCos = 2 * cos.
DEV float FrConductorBin(float cos2, float sin2, float twoCos, float etaT, float k) {
    const float eta = etaT / 1.f, etak = k / 1.f;
    const float eta2 = eta * eta, etak2 = etak * etak;
    const float t0 = (eta2 - etak2) - sin2;
    const float a2plusb2 = __builtin_sqrtf(t0 * t0 + (eta2 * 4.f) * etak2);
    const float t1 = a2plusb2 + cos2;
    const float a = __builtin_sqrtf((a2plusb2 + t0) * 0.5f);
    const float t2 = a * twoCos;
    const float Rs = (t1 - t2) / (t1 + t2);
    const float t3 = a2plusb2 * cos2 + sin2 * sin2;
    const float t4 = t2 * sin2;
    const float Rp = (Rs * (t3 - t4)) / (t3 + t4);
    return (Rp + Rs) * 0.5f;
}

// Spectrum::FromRGB(rgb, type) (SampledSpectrum::FromRGB, spectrum.cpp:98-180) reduced to scalars:
// bin value = Clamp(((0 + white*w0) + basis[i1]*w1) + basis[i2]*w2) * .86445f, 0, inf).
struct IllumRGB {
    int i1, i2;
    float w0, w1, w2;
};
DEV IllumRGB MakeIllumRGB(const float rgb[3]) {
    IllumRGB q;
    if (rgb[0] <= rgb[1] && rgb[0] <= rgb[2]) {
        q.w0 = rgb[0];
        if (rgb[1] <= rgb[2]) { q.i1 = 1; q.w1 = rgb[1] - rgb[0]; q.i2 = 6; q.w2 = rgb[2] - rgb[1]; }
        else { q.i1 = 1; q.w1 = rgb[2] - rgb[0]; q.i2 = 5; q.w2 = rgb[1] - rgb[2]; }
    } else if (rgb[1] <= rgb[0] && rgb[1] <= rgb[2]) {
        q.w0 = rgb[1];
        if (rgb[0] <= rgb[2]) { q.i1 = 2; q.w1 = rgb[0] - rgb[1]; q.i2 = 6; q.w2 = rgb[2] - rgb[0]; }
        else { q.i1 = 2; q.w1 = rgb[2] - rgb[1]; q.i2 = 4; q.w2 = rgb[0] - rgb[2]; }
    } else {
        q.w0 = rgb[2];
        if (rgb[0] <= rgb[1]) { q.i1 = 3; q.w1 = rgb[0] - rgb[2]; q.i2 = 5; q.w2 = rgb[1] - rgb[0]; }
        else { q.i1 = 3; q.w1 = rgb[1] - rgb[2]; q.i2 = 4; q.w2 = rgb[0] - rgb[1]; }
    }
    return q;
}

// Image-textured lobes of the material at this vertex (mi_lobe_tex): per material lobe the texture value in FromRGB's
// compact form; bit i of hasR / hasS: lobe i takes R / S from it, of mulR / mulS: multiplied onto the constant.
template <int NL>
struct LobeTexT {
    unsigned hasR, hasS, mulR, mulS;
    unsigned hasK;    // bit i: lobe i ("metal", MI_LOBE_METAL) takes its k from the texture value kept in r[i] -- its R stays the constant
    unsigned rules;   // mi_lobe_rule of lobe i at bits 4i..4i+3 (the "disney" rules derive a lobe's spectrum from the colour; read by
                      // the eight-lobe instances only: a Disney material has more than four lobes, and inlined into every spectral
                      // access of the two- and four-lobe instances the rule test cost the textured zoo 3.5 % of its shading time)
    float lum;        // "disney" with a textured colour: c.y() of the colour at this vertex
    IllumRGB r[NL], s[NL];
    const float *basis;     // rgbIllum2Spect{White..Blue}, [7][31]: FromRGB's default type is Illuminant (spectrum.h:428-429)
    const mi_texture *textures;   // for checkerboard values (i1 < 0: i2 = texture, w0 = weight of its spec2)
};
// Bin of Spectrum::FromRGB(rgb).Clamp(): Clamp((((0 + white*w0) + basis[i1]*w1) + basis[i2]*w2) * .86445f, 0, inf)
DEV float TexBin(const float *basis, const mi_texture *textures, const IllumRGB &q, int bin) {
    if (q.i1 < 0) {   // Checkerboard2DTexture: (1 - area2) * tex1 + area2 * tex2, then the material's Clamp()
        const mi_texture &t = textures[q.i2];
        return clampf((1 - q.w0) * t.spec1[bin] + q.w0 * t.spec2[bin], 0.f, kInfinity);
    }
    float r = 0.f;
    r += basis[bin] * q.w0;
    r += basis[q.i1 * MI_NSPEC + bin] * q.w1;
    r += basis[q.i2 * MI_NSPEC + bin] * q.w2;
    r *= .86445f;
    return clampf(r, 0.f, kInfinity);
}
// "disney" with an image-textured colour (disney.cpp:485-587; mi_lobe_rule in mi_pt.h): the channel of a lobe whose spectrum is
// not linear in the colour, from the colour's bin c and luminance lum. Spectrum arithmetic's order: Lerp(t, a, b) = a * (1 - t)
// + b * t (spectrum.h:577-580), Float * Spectrum = the bin times the float.
DEV float DisneyTexBin(int rule, int which, const float *p, float c, float lum) {
    const float ctint = lum > 0 ? c / lum : 1.f;                                  // Ctint = lum > 0 ? c / lum : Spectrum(1.)
    if (rule == MI_LOBE_DISNEY_SHEEN) return ((1 - p[7]) + ctint * p[7]) * p[6];  // Lerp(sheenTint, 1, Ctint) * (diffuseWeight * sheenWeight)
    if (rule == MI_LOBE_DISNEY_STRANS) return __builtin_sqrtf(c) * p[6];          // Sqrt(c) * strans
    if (which == 0) return c;                                                      // MI_LOBE_DISNEY_SPEC: R = c,
    const float x = ((1 - p[6]) + ctint * p[6]) * p[7];                            //   S = Lerp(metallic, Lerp(specTint, 1, Ctint) * R0, c)
    return x * (1 - p[2]) + c * p[2];
}
// R (which = 0) or S (which = 1) of material lobe li at a bin, with the texture applied
template <int NL>
DEV float TexturedSpec(const LobeTexT<NL> &lt, const mi_bxdf &b, int li, int which, int bin) {
    const float c = which ? b.S[bin] : b.R[bin];
    if (!(((which ? lt.hasS : lt.hasR) >> li) & 1u)) return c;
    const float T = TexBin(lt.basis, lt.textures, which ? lt.s[li] : lt.r[li], bin);
    if constexpr (NL == MI_MAX_BXDFS) {
        const int rule = (int)((lt.rules >> (4 * li)) & 15u);
        if (rule >= MI_LOBE_DISNEY_SHEEN && rule <= MI_LOBE_DISNEY_STRANS) return DisneyTexBin(rule, which, b.p, T, lt.lum);
    }
    return (((which ? lt.mulS : lt.mulR) >> li) & 1u) ? c * T : T;
}
// The conductor's absorption k at a bin: "metal" with an image-textured k keeps that texture in the lobe's R slot (MI_LOBE_METAL)
template <int NL>
DEV float TexturedK(const LobeTexT<NL> &lt, const mi_bxdf &b, int li, int bin) {
    if ((lt.hasK >> li) & 1u) return TexBin(lt.basis, lt.textures, lt.r[li], bin);
    return b.K[bin];
}

// Lobe-type masks: the shading kernel is instantiated for sets of BxDF types (bit = mi_bxdf_type, bits 16.. =
// mi_fresnel_type), so that a wave shading matte surfaces carries no microfacet or Disney code and registers.
// The guards compile the other cases out; a lobe outside the mask cannot occur (the host picks the kernel from
// the lobes the class holds).
#define TM_HAS(tm, type) ((((tm) >> (type)) & 1u) != 0u)
#define TM_FRESNEL(tm, f) ((((tm) >> (16 + (f))) & 1u) != 0u)
#define TM_SPECULAR(tm) (TM_HAS(tm, MI_BXDF_SPECULAR_REFLECTION) || TM_HAS(tm, MI_BXDF_SPECULAR_TRANSMISSION) || TM_HAS(tm, MI_BXDF_FRESNEL_SPECULAR))
constexpr unsigned TM_ALL = 0xffffffffu;
constexpr unsigned TM_SCALED = 1u << 31;
constexpr unsigned TM_SAMPLERS = 1u << 15;   // the instance draws from the Sobol' and random samplers too (without it: Halton only)
constexpr unsigned TM_INSTANCES = 1u << 29;  // the scene has object instances: hits carry the instance they were reached through
constexpr unsigned TM_TEXTURED = 1u << 30;  // some lobe takes its spectrum from an image texture  // some lobe is a ScaledBxDF (mix material)
#define TM_LIGHT(tm, t) ((((tm) >> (24 + (t))) & 1u) != 0u)   // bits 24..: mi_light_type present in the scene
constexpr unsigned TM_LIGHTS_ALL = 0x1fu << 24;
constexpr unsigned TM_LIGHTS_NO_ENV = TM_LIGHTS_ALL & ~(1u << (24 + MI_LIGHT_INFINITE));
constexpr unsigned TM_DIFFUSE = (1u << MI_BXDF_LAMBERTIAN_REFLECTION) | (1u << MI_BXDF_OREN_NAYAR) | (1u << (16 + MI_FRESNEL_NOOP));
constexpr unsigned TM_PLASTIC = TM_DIFFUSE | (1u << MI_BXDF_MICROFACET_REFLECTION) | (1u << (16 + MI_FRESNEL_DIELECTRIC));
// the lobes "glass" and "mirror" make (glass.cpp:60-92, mirror.cpp:46-56): specular and rough dielectric interfaces
constexpr unsigned TM_GLASS = (1u << MI_BXDF_SPECULAR_REFLECTION) | (1u << MI_BXDF_SPECULAR_TRANSMISSION) | (1u << MI_BXDF_FRESNEL_SPECULAR) |
                              (1u << MI_BXDF_MICROFACET_REFLECTION) | (1u << MI_BXDF_MICROFACET_TRANSMISSION) |
                              (1u << (16 + MI_FRESNEL_DIELECTRIC)) | (1u << (16 + MI_FRESNEL_NOOP));
// the lobes "uber" makes (uber.cpp:60-105; "translucent"'s fit as well) and the ones "disney" makes (disney.cpp:474-587)
constexpr unsigned TM_UBER = TM_PLASTIC | (1u << MI_BXDF_SPECULAR_REFLECTION) | (1u << MI_BXDF_SPECULAR_TRANSMISSION) |
                             (1u << MI_BXDF_MICROFACET_TRANSMISSION) | (1u << MI_BXDF_LAMBERTIAN_TRANSMISSION);
constexpr unsigned TM_DISNEY = (1u << MI_BXDF_DISNEY_DIFFUSE) | (1u << MI_BXDF_DISNEY_FAKE_SS) | (1u << MI_BXDF_DISNEY_RETRO) | (1u << MI_BXDF_DISNEY_SHEEN) |
                               (1u << MI_BXDF_DISNEY_CLEARCOAT) | (1u << MI_BXDF_MICROFACET_REFLECTION) | (1u << MI_BXDF_MICROFACET_TRANSMISSION) |
                               (1u << MI_BXDF_LAMBERTIAN_TRANSMISSION) | (1u << MI_BXDF_SPECULAR_TRANSMISSION) | (1u << MI_BXDF_LAMBERTIAN_REFLECTION) |
                               (1u << (16 + MI_FRESNEL_DISNEY)) | (1u << (16 + MI_FRESNEL_DIELECTRIC)) | (1u << (16 + MI_FRESNEL_NOOP));

template <unsigned TM>
DEV float LobeValueCore(const LobeEval &le, float R, float Sv, float Kv) {  // R: the lobe's spectrum at the bin (S when bit 8 of le.lobe is set)
    switch (le.kind & 0xff) {
    case LK_MICRO_CONDUCTOR: if constexpr (TM_FRESNEL(TM, MI_FRESNEL_CONDUCTOR)) {
        const float F = FrConductorBin(le.c, le.e, le.f, Sv, Kv);
        return LobeDiv(((R * le.a) * le.b) * F, le);
    } break;
    case LK_FBLEND: if constexpr (TM_HAS(TM, MI_BXDF_FRESNEL_BLEND)) {
        const float Rs = Sv;
        const float diffuse = (((R * le.f) * (1.f - Rs)) * le.a) * le.b;
        const float specular = (Rs + (1.f - Rs) * le.c) * le.e;
        return diffuse + specular;
    } break;
    case LK_MUL1: return R * le.a;
    case LK_MUL2: return (R * le.a) * le.b;
    case LK_MUL3: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_DIFFUSE) || TM_HAS(TM, MI_BXDF_DISNEY_RETRO)) return ((R * le.a) * le.b) * le.c; break;
    case LK_MUL3_DIV: if constexpr (TM_HAS(TM, MI_BXDF_MICROFACET_REFLECTION)) return LobeDiv(((R * le.a) * le.b) * le.c, le); break;
    case LK_MUL1_DIV: if constexpr (TM_SPECULAR(TM)) return LobeDiv(R * le.a, le); break;
    case LK_MUL2_DIV: if constexpr (TM_SPECULAR(TM)) return LobeDiv((R * le.a) * le.b, le); break;
    case LK_MICRO_DISNEY: if constexpr (TM_FRESNEL(TM, MI_FRESNEL_DISNEY)) {
        float S = Sv;
        // Lerp(metallic, Spectrum(FrDielectric), FrSchlick(R0, cosI)); FrSchlick = Lerp(w, R0, 1)
        float schlick = (1 - le.f) * S + le.f * 1.f;
        float F = (1 - le.c) * le.e + le.c * schlick;
        return LobeDiv(((R * le.a) * le.b) * F, le);
    } break;
    case LK_MTRANS: if constexpr (TM_HAS(TM, MI_BXDF_MICROFACET_TRANSMISSION)) return ((1.f - le.a) * R) * le.b;
    case LK_CONST: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_CLEARCOAT)) return le.a;
    default: break;
    }
    return 0.f;
}
// Which of a lobe's other spectra its value reads (compile-time, from the kernel's lobe mask).
#define TM_NEEDS_S(tm) (TM_FRESNEL(tm, MI_FRESNEL_CONDUCTOR) || TM_FRESNEL(tm, MI_FRESNEL_DISNEY) || TM_HAS(tm, MI_BXDF_FRESNEL_BLEND))
#define TM_NEEDS_K(tm) (TM_FRESNEL(tm, MI_FRESNEL_CONDUCTOR))
template <int NL, unsigned TM>
DEV float LobeValueInner(const LobeEval &le, const mi_bxdf *bx, int bin, const LobeTexT<NL> *lt) {
    const int li = le.lobe & 0xff;  // bit 8 set: the lobe's second spectrum (T of FresnelSpecular)
    float R, Sv = 0.f, Kv = 0.f;
    if constexpr ((TM & TM_TEXTURED) != 0) {
        R = TexturedSpec(*lt, bx[li], li, (le.lobe & 0x100) ? 1 : 0, bin);
        if constexpr (TM_NEEDS_S(TM)) Sv = TexturedSpec(*lt, bx[li], li, 1, bin);
    } else {
        R = (le.lobe & 0x100) ? bx[li].S[bin] : bx[li].R[bin];
        if constexpr (TM_NEEDS_S(TM)) Sv = bx[li].S[bin];
    }
    if constexpr (TM_NEEDS_K(TM)) {
        if constexpr ((TM & TM_TEXTURED) != 0) Kv = TexturedK(*lt, bx[li], li, bin);
        else Kv = bx[li].K[bin];
    }
    return LobeValueCore<TM>(le, R, Sv, Kv);
}
// bit 9 of le.lobe: the lobe is wrapped in a ScaledBxDF (mix material), f = scale * f (reflection.cpp:96-107); bit 10: in two of
// them (a "mix" of a "mix"), f = scale2 * (scale * f)
template <int NL, unsigned TM>
DEV float LobeValue(const LobeEval &le, const mi_bxdf *bx, int bin, const LobeTexT<NL> *lt) {
    const float v = LobeValueInner<NL, TM>(le, bx, bin, lt);
    if constexpr ((TM & TM_SCALED) == 0) return v;
    if (!(le.lobe & 0x200)) return v;
    const float v1 = bx[le.lobe & 0xff].scale[bin] * v;
    return (le.lobe & 0x400) ? bx[le.lobe & 0xff].scale2[bin] * v1 : v1;
}

// Microfacet roughness from float textures ("texture roughness" / "uroughness" / "vroughness": plastic.cpp:57-62,
// uber.cpp:88-96, substrate.cpp:55-60, ...): the alphas of the material's microfacet lobes at this vertex, in place of the
// constants in mi_bxdf.p[0..1] (mi_material.rough_tex; every such material gives all its microfacet lobes the same pair).
struct AlphaOv {
    float u, v;
    bool onU, onV;
    // "matte" with `sigma` a float image texture (mi_material.sigma_tex; matte.cpp:55-62): 0 = the lobe's constants, 1 = OrenNayar
    // with sigA / sigB (the A, B of the sigma at this vertex, reflection.h:414-420), 2 = sigma is 0 here: LambertianReflection
    int sigMode;
    float sigA, sigB;
    // "disney" with `roughness` a float image texture (MI_ROUGH_DISNEY): the value at this vertex -- p[0] of the FakeSS / Retro lobes,
    // and the source of the microfacet lobes' alphas (DistOf)
    bool disney;
    float rough;
};
// TrowbridgeReitzDistribution::RoughnessToAlpha, microfacet.h:140-145
DEV float RoughnessToAlpha(float roughness) {
    roughness = maxf(roughness, 1e-3f);
    const float x = logF(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
struct BSDFFrame {
    V3 ns, ng, ss, ts;
    const mi_material *m;
    AlphaOv ov;      // (read by the texture-evaluating instances only)
    unsigned mask;   // bit i: lobe i of m is part of the BSDF at this vertex (textured lobes drop out where their texture is black)
    DEV bool On(int i) const { return ((mask >> i) & 1u) != 0u; }
    DEV V3 WorldToLocal(const V3 &v) const { return V3(Dot(v, ss), Dot(v, ts), Dot(v, ns)); }
    DEV V3 LocalToWorld(const V3 &v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z,
                  ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
};
DEV bool MatchesFlags(const mi_bxdf &b, int t) { return (b.flags & t) == b.flags; }
DEV int NumComponents(const BSDFFrame &fr, int flags) {
    const mi_material *m = fr.m;
    int num = 0;
    for (int i = 0; i < m->n_bxdfs; ++i) if (fr.On(i) && MatchesFlags(m->bxdf[i], flags)) ++num;
    return num;
}
template <unsigned TM>
DEV TRDist DistOf(const mi_bxdf &b, const AlphaOv &ov) {
    if constexpr ((TM & TM_TEXTURED) != 0) {
        if (ov.disney) {   // disney.cpp:538-541, 568-573: p[4] = aspect, p[7] = 0.65 eta - 0.35 on the thin surface's transmission lobe
            const float r = (b.type == MI_BXDF_MICROFACET_TRANSMISSION && b.p[7] != 0.f) ? b.p[7] * ov.rough : ov.rough;
            return TRDist{maxf(.001f, (r * r) / b.p[4]), maxf(.001f, (r * r) * b.p[4]), b.p[5] != 0.f};
        }
        return TRDist{ov.onU ? ov.u : b.p[0], ov.onV ? ov.v : b.p[1], b.p[5] != 0.f};
    }
    return TRDist{b.p[0], b.p[1], b.p[5] != 0.f};
}

// BxDF::f for lobe i (local wo, wi) -> LobeEval. Mirrors o_bsdf / reflection.cpp per lobe.
template <unsigned TM>
DEV LobeEval LobeF(const mi_bxdf &b, int i, const V3 &wo, const V3 &wi, const AlphaOv &ov) {
    LobeEval le;
    le.kind = LK_NONE; le.lobe = i | (b.scaled ? 0x200 : 0) | (b.scaled >= 2 ? 0x400 : 0); le.a = le.b = le.c = le.d = le.e = le.f = le.r = 0;
    switch (b.type) {
    case MI_BXDF_FRESNEL_BLEND: if constexpr (TM_HAS(TM, MI_BXDF_FRESNEL_BLEND)) {  // reflection.cpp:285-298
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) break;
        wh = Normalize(wh);
        const float hi = 1 - .5f * AbsCosTheta(wi), ho = 1 - .5f * AbsCosTheta(wo), hc = 1 - Dot(wi, wh);
        le.kind = LK_FBLEND;
        le.f = (28.f / (23.f * kPi));
        le.a = (1 - (hi * hi) * (hi * hi) * hi);
        le.b = (1 - (ho * ho) * (ho * ho) * ho);
        le.c = (hc * hc) * (hc * hc) * hc;
        le.e = DistOf<TM>(b, ov).D(wh) / (4 * AbsDot(wi, wh) * maxf(AbsCosTheta(wi), AbsCosTheta(wo)));
        break;
    } break;
    case MI_BXDF_LAMBERTIAN_REFLECTION:
    case MI_BXDF_LAMBERTIAN_TRANSMISSION:
        le.kind = LK_MUL1; le.a = kInvPi; break;
    case MI_BXDF_OREN_NAYAR: if constexpr (TM_HAS(TM, MI_BXDF_OREN_NAYAR)) {
        if constexpr ((TM & TM_TEXTURED) != 0) { if (ov.sigMode == 2) { le.kind = LK_MUL1; le.a = kInvPi; break; } }
        float sinThetaI = SinTheta(wi), sinThetaO = SinTheta(wo);
        float maxCos = 0;
        if ((double)sinThetaI > 1e-4 && (double)sinThetaO > 1e-4) {
            float sinPhiI = SinPhi(wi), cosPhiI = CosPhi(wi);
            float sinPhiO = SinPhi(wo), cosPhiO = CosPhi(wo);
            float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
            maxCos = maxf(0.f, dCos);
        }
        float sinAlpha, tanBeta;
        if (AbsCosTheta(wi) > AbsCosTheta(wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / AbsCosTheta(wi); }
        else { sinAlpha = sinThetaI; tanBeta = sinThetaO / AbsCosTheta(wo); }
        float oA = b.p[0], oB = b.p[1];
        if constexpr ((TM & TM_TEXTURED) != 0) { if (ov.sigMode == 1) { oA = ov.sigA; oB = ov.sigB; } }
        le.kind = LK_MUL2; le.a = kInvPi; le.b = (oA + oB * maxCos * sinAlpha * tanBeta);
        break;
    } break;
    case MI_BXDF_MICROFACET_REFLECTION: if constexpr (TM_HAS(TM, MI_BXDF_MICROFACET_REFLECTION)) {
        float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
        V3 wh = wi + wo;
        if (cosThetaI == 0 || cosThetaO == 0) break;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) break;
        wh = Normalize(wh);
        TRDist d = DistOf<TM>(b, ov);
        float cosI = Dot(wi, wh);
        le.a = d.D(wh); le.b = d.G(wo, wi);
        const float denom = (4 * cosThetaI * cosThetaO);
        if (TM_FRESNEL(TM, MI_FRESNEL_CONDUCTOR) && b.fresnel == MI_FRESNEL_CONDUCTOR) {  // FresnelConductor::Evaluate(cosI) = FrConductor(|cosI|, 1, eta, k)
            le.kind = LK_MICRO_CONDUCTOR;
            const float c = clampf(absf(cosI), -1, 1);
            le.c = c * c;
            le.e = (float)(1. - (double)le.c);
            le.f = (float)2 * c;
        } else if (TM_FRESNEL(TM, MI_FRESNEL_DISNEY) && b.fresnel == MI_FRESNEL_DISNEY) {
            le.kind = LK_MICRO_DISNEY;
            le.c = b.p[2];                       // metallic
            le.e = FrDielectric(cosI, 1, b.p[3]);
            le.f = SchlickWeight(cosI);
        } else {
            le.kind = LK_MUL3_DIV;
            le.c = (b.fresnel == MI_FRESNEL_DIELECTRIC) ? FrDielectric(cosI, b.p[2], b.p[3]) : 1.f;
        }
        SetDivisor(le, denom);
        break;
    } break;
    case MI_BXDF_MICROFACET_TRANSMISSION: if constexpr (TM_HAS(TM, MI_BXDF_MICROFACET_TRANSMISSION)) {
        if (SameHemisphere(wo, wi)) break;
        float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
        if (cosThetaI == 0 || cosThetaO == 0) break;
        const float etaA = b.p[2], etaB = b.p[3];
        float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
        V3 wh = Normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        float F = FrDielectric(Dot(wo, wh), etaA, etaB);
        float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
        float factor = 1 / eta;
        TRDist d = DistOf<TM>(b, ov);
        le.kind = LK_MTRANS;
        le.a = F;
        le.b = absf(d.D(wh) * d.G(wo, wi) * eta * eta * AbsDot(wi, wh) * AbsDot(wo, wh) * factor * factor /
                    (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
        break;
    } break;
    case MI_BXDF_DISNEY_DIFFUSE: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_DIFFUSE)) {
        float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
        le.kind = LK_MUL3; le.a = kInvPi; le.b = (1 - Fo / 2); le.c = (1 - Fi / 2);
        break;
    } break;
    case MI_BXDF_DISNEY_FAKE_SS: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_FAKE_SS)) {
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) break;
        wh = Normalize(wh);
        float cosThetaD = Dot(wi, wh);
        float Fss90 = cosThetaD * cosThetaD * (((TM & TM_TEXTURED) != 0 && ov.disney) ? ov.rough : b.p[0]);
        float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
        float Fss = lerpf(Fo, 1.0f, Fss90) * lerpf(Fi, 1.0f, Fss90);
        float ss = 1.25f * (Fss * (1 / (AbsCosTheta(wo) + AbsCosTheta(wi)) - .5f) + .5f);
        le.kind = LK_MUL2; le.a = kInvPi; le.b = ss;
        break;
    } break;
    case MI_BXDF_DISNEY_RETRO: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_RETRO)) {
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) break;
        wh = Normalize(wh);
        float cosThetaD = Dot(wi, wh);
        float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
        float Rr = 2 * (((TM & TM_TEXTURED) != 0 && ov.disney) ? ov.rough : b.p[0]) * cosThetaD * cosThetaD;
        le.kind = LK_MUL3; le.a = kInvPi; le.b = Rr; le.c = (Fo + Fi + Fo * Fi * (Rr - 1));
        break;
    } break;
    case MI_BXDF_DISNEY_SHEEN: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_SHEEN)) {
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) break;
        wh = Normalize(wh);
        le.kind = LK_MUL1; le.a = SchlickWeight(Dot(wi, wh));
        break;
    } break;
    case MI_BXDF_DISNEY_CLEARCOAT: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_CLEARCOAT)) {
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) break;
        wh = Normalize(wh);
        float Dr = GTR1(AbsCosTheta(wh), b.p[1]);
        float Fr = FrSchlickF(.04f, Dot(wo, wh));
        float Gr = smithG_GGX(AbsCosTheta(wo), .25f) * smithG_GGX(AbsCosTheta(wi), .25f);
        le.kind = LK_CONST; le.a = b.p[0] * Gr * Fr * Dr / 4;
        break;
    } break;
    default: break;  // specular lobes: f == 0
    }
    return le;
}

template <unsigned TM>
DEV float LobePdf(const mi_bxdf &b, const V3 &wo, const V3 &wi, const AlphaOv &ov) {
    switch (b.type) {
    case MI_BXDF_FRESNEL_BLEND: if constexpr (TM_HAS(TM, MI_BXDF_FRESNEL_BLEND)) {  // reflection.cpp:470-475
        if (!SameHemisphere(wo, wi)) return 0;
        V3 wh = Normalize(wo + wi);
        float pdf_wh = DistOf<TM>(b, ov).Pdf(wo, wh);
        return .5f * (AbsCosTheta(wi) * kInvPi + pdf_wh / (4 * Dot(wo, wh)));
    } break;
    case MI_BXDF_SPECULAR_REFLECTION: case MI_BXDF_SPECULAR_TRANSMISSION: case MI_BXDF_FRESNEL_SPECULAR: if constexpr (TM_HAS(TM, MI_BXDF_SPECULAR_REFLECTION) || TM_HAS(TM, MI_BXDF_SPECULAR_TRANSMISSION) || TM_HAS(TM, MI_BXDF_FRESNEL_SPECULAR)) { return 0; } break;
    case MI_BXDF_LAMBERTIAN_TRANSMISSION: if constexpr (TM_HAS(TM, MI_BXDF_LAMBERTIAN_TRANSMISSION)) { return !SameHemisphere(wo, wi) ? AbsCosTheta(wi) * kInvPi : 0; } break;
    case MI_BXDF_MICROFACET_REFLECTION: if constexpr (TM_HAS(TM, MI_BXDF_MICROFACET_REFLECTION)) {
        if (!SameHemisphere(wo, wi)) return 0;
        V3 wh = Normalize(wo + wi);
        return DistOf<TM>(b, ov).Pdf(wo, wh) / (4 * Dot(wo, wh));
    } break;
    case MI_BXDF_MICROFACET_TRANSMISSION: if constexpr (TM_HAS(TM, MI_BXDF_MICROFACET_TRANSMISSION)) {
        if (SameHemisphere(wo, wi)) return 0;
        const float etaA = b.p[2], etaB = b.p[3];
        float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
        V3 wh = Normalize(wo + wi * eta);
        float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
        float dwh_dwi = absf((eta * eta * Dot(wi, wh)) / (sqrtDenom * sqrtDenom));
        return DistOf<TM>(b, ov).Pdf(wo, wh) * dwh_dwi;
    } break;
    case MI_BXDF_DISNEY_CLEARCOAT: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_CLEARCOAT)) {
        if (!SameHemisphere(wo, wi)) return 0;
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return 0;
        wh = Normalize(wh);
        float Dr = GTR1(AbsCosTheta(wh), b.p[1]);
        return Dr * AbsCosTheta(wh) / (4 * Dot(wo, wh));
    } break;
    default: break;
    }
    return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * kInvPi : 0;
}

// NL = compile-time bound on the material's lobe count: the shading kernels are
// instantiated per material class (<= 2 lobes: matte / plastic / glass / mirror; up to 8:
// uber / disney) so the common class keeps its lobe list in 16 registers.
template <int NL>
struct BSDFEvalT {
    int n;  // number of contributing lobes
    LobeEval lobes[NL];
};

// BSDF::f(woW, wiW, flags): fills the lobe list (reflection.cpp:670-683).
template <int NL, unsigned TM>
DEV void BSDF_f(const BSDFFrame &fr, const V3 &woW, const V3 &wiW, int flags, BSDFEvalT<NL> *ev) {
    ev->n = 0;
    V3 wi = fr.WorldToLocal(wiW), wo = fr.WorldToLocal(woW);
    if (wo.z == 0) return;
    bool reflect = Dot(wiW, fr.ng) * Dot(woW, fr.ng) > 0;
    const mi_material *m = fr.m;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        if (i < m->n_bxdfs && fr.On(i)) {
            const mi_bxdf &b = m->bxdf[i];
            if (MatchesFlags(b, flags) &&
                ((reflect && (b.flags & MI_BSDF_REFLECTION)) || (!reflect && (b.flags & MI_BSDF_TRANSMISSION)))) {
                LobeEval le = LobeF<TM>(b, i, wo, wi, fr.ov);
                if ((le.kind & 0xff) != LK_NONE) ev->lobes[ev->n++] = le;
            }
        }
    }
}
template <unsigned TM>
DEV float BSDF_Pdf(const BSDFFrame &fr, const V3 &woW, const V3 &wiW, int flags) {  // reflection.cpp:770-785
    const mi_material *m = fr.m;
    if (m->n_bxdfs == 0 || (fr.mask & ((1u << m->n_bxdfs) - 1u)) == 0u) return 0.f;
    V3 wo = fr.WorldToLocal(woW), wi = fr.WorldToLocal(wiW);
    if (wo.z == 0) return 0.;
    float pdf = 0.f;
    int matchingComps = 0;
    for (int i = 0; i < m->n_bxdfs; ++i)
        if (fr.On(i) && MatchesFlags(m->bxdf[i], flags)) { ++matchingComps; pdf += LobePdf<TM>(m->bxdf[i], wo, wi, fr.ov); }
    return matchingComps > 0 ? pdf / matchingComps : 0.f;
}

template <int NL, unsigned TM>
DEV float EvalBin(const BSDFEvalT<NL> &ev, const mi_bxdf *bx, int bin, const LobeTexT<NL> *lt) {
    float f = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i)
        if (i < ev.n) f += LobeValue<NL, TM>(ev.lobes[i], bx, bin, lt);
    return f;
}

// Four bins (quad c) of the lobe list at once. The lobes' spectra are fetched first, as unaligned 16-B loads that
// do not depend on the lobe kind, so that one quad costs one memory round trip; fetched bin by bin inside the kind
// switch, every bin of every lobe waits for its own load. Bin 31 (the pad lane of quad 7) reads the 4 bytes after
// the array -- still inside the material table, whose allocation carries 16 B of slack -- and is never used.
typedef float __attribute__((ext_vector_type(4), aligned(4))) float4u;
DEV float4 LoadSpec4(const float *spec, int c) {
    const float4u v = *reinterpret_cast<const float4u *>(spec + 4 * c);
    return make_float4(v.x, v.y, v.z, v.w);
}
DEV float Quad(const float4 &v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
// TexBin / TexturedSpec for the four bins of quad c at once: the basis tables and the lobe's constant as 16-B loads (a textured
// pass fetched four words per bin and lobe: ~750 loads per vertex on the textured zoo). Same operations per bin.
DEV float4 TexQuad(const float *basis, const mi_texture *textures, const IllumRGB &q, int c) {
    if (q.i1 < 0) {   // Checkerboard2DTexture
        const mi_texture &t = textures[q.i2];
        const float4 a = LoadSpec4(t.spec1, c), b = LoadSpec4(t.spec2, c);
        return make_float4(clampf((1 - q.w0) * a.x + q.w0 * b.x, 0.f, kInfinity), clampf((1 - q.w0) * a.y + q.w0 * b.y, 0.f, kInfinity),
                           clampf((1 - q.w0) * a.z + q.w0 * b.z, 0.f, kInfinity), clampf((1 - q.w0) * a.w + q.w0 * b.w, 0.f, kInfinity));
    }
    const float4 b0 = LoadSpec4(basis, c), b1 = LoadSpec4(basis + q.i1 * MI_NSPEC, c), b2 = LoadSpec4(basis + q.i2 * MI_NSPEC, c);
    auto one = [&](float x0, float x1, float x2) {
        float r = 0.f;
        r += x0 * q.w0;
        r += x1 * q.w1;
        r += x2 * q.w2;
        r *= .86445f;
        return clampf(r, 0.f, kInfinity);
    };
    return make_float4(one(b0.x, b1.x, b2.x), one(b0.y, b1.y, b2.y), one(b0.z, b1.z, b2.z), one(b0.w, b1.w, b2.w));
}
template <int NL>
DEV float4 TexturedQuad(const LobeTexT<NL> &lt, const mi_bxdf &b, int li, int which, int c) {
    const float4 k = LoadSpec4(which ? b.S : b.R, c);
    if (!(((which ? lt.hasS : lt.hasR) >> li) & 1u)) return k;
    const float4 T = TexQuad(lt.basis, lt.textures, which ? lt.s[li] : lt.r[li], c);
    if constexpr (NL == MI_MAX_BXDFS) {
        const int rule = (int)((lt.rules >> (4 * li)) & 15u);
        if (rule >= MI_LOBE_DISNEY_SHEEN && rule <= MI_LOBE_DISNEY_STRANS)
            return make_float4(DisneyTexBin(rule, which, b.p, T.x, lt.lum), DisneyTexBin(rule, which, b.p, T.y, lt.lum),
                               DisneyTexBin(rule, which, b.p, T.z, lt.lum), DisneyTexBin(rule, which, b.p, T.w, lt.lum));
    }
    if (((which ? lt.mulS : lt.mulR) >> li) & 1u) return make_float4(k.x * T.x, k.y * T.y, k.z * T.z, k.w * T.w);
    return T;
}
template <int NL, unsigned TM>
DEV float4 EvalQuad(const BSDFEvalT<NL> &ev, const mi_bxdf *bx, int c, const LobeTexT<NL> *lt) {
    if constexpr (NL > 2 || (TM & TM_TEXTURED) != 0) {   // long lobe lists: the quads of all lobes would not fit the register file
        const int b = 4 * c;
        return make_float4(EvalBin<NL, TM>(ev, bx, b, lt), EvalBin<NL, TM>(ev, bx, b + 1, lt), EvalBin<NL, TM>(ev, bx, b + 2, lt),
                           (b + 3 < MI_NSPEC) ? EvalBin<NL, TM>(ev, bx, b + 3, lt) : 0.f);
    }
    float4 R[NL], Sv[NL], Kv[NL], Sc[NL], Sc2[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        R[i] = Sv[i] = Kv[i] = Sc[i] = Sc2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < ev.n) {
            const int lobe = ev.lobes[i].lobe, li = lobe & 0xff;
            R[i] = LoadSpec4((lobe & 0x100) ? bx[li].S : bx[li].R, c);
            if constexpr (TM_NEEDS_S(TM)) Sv[i] = LoadSpec4(bx[li].S, c);
            if constexpr (TM_NEEDS_K(TM)) Kv[i] = LoadSpec4(bx[li].K, c);
            if constexpr ((TM & TM_SCALED) != 0) { if (lobe & 0x200) Sc[i] = LoadSpec4(bx[li].scale, c); if (lobe & 0x400) Sc2[i] = LoadSpec4(bx[li].scale2, c); }
        }
    }
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = 0.f;
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < ev.n) {
                float v = LobeValueCore<TM>(ev.lobes[i], Quad(R[i], k), Quad(Sv[i], k), Quad(Kv[i], k));
                if constexpr ((TM & TM_SCALED) != 0) { if (ev.lobes[i].lobe & 0x200) v = Quad(Sc[i], k) * v; if (ev.lobes[i].lobe & 0x400) v = Quad(Sc2[i], k) * v; }
                f[k] += v;
            }
    }
    return make_float4(f[0], f[1], f[2], f[3]);
}

// ---- straight-line spectral evaluation (the hot shading instances: matte, plastic, glass / mirror lobe sets)
// EvalQuad / LobeValueCore / DivBy decide per lane and per bin -- which kind of lobe, whether the fast quotient applies,
// whether bin 31 exists -- and every such decision is an exec-mask branch: the light-sample loop of the plastic instance was
// 815 instructions per quad, 380 of them branch scaffolding (s_and_saveexec / s_cbranch_execz / s_or), in a kernel bound by
// instruction issue (SQ counters: VALU + SALU issue fill the four waves' slots). Here the same values come out of code
// without a per-lane branch:
//  * every lobe kind these instances can meet is ((R*a)*b)*c [/ d] with factors that may be absent; an absent factor is 1
//    (x * 1.f == x exactly), LK_MTRANS's (1 - a) * R is R * (1 - a) (IEEE multiplication commutes); whether ANY lane of the
//    wave needs the b / c factor or the division is a wave-uniform (scalar) branch;
//  * DivBy's quotient candidate is formed for every operand without testing it; instead the largest and smallest magnitude
//    seen in the quad are tracked as integers (two instructions per quotient) and ONE wave-uniform test per quad decides
//    whether any lane left DivBy's fast range -- then the quad is redone by the branching code (EvalQuad / DivBy), which
//    is the definition of the result. (An exact zero stays on the fast path: 0 * r is the quotient 0 / d.)
#define TM_SIMPLE_KINDS(tm) (!TM_FRESNEL(tm, MI_FRESNEL_CONDUCTOR) && !TM_FRESNEL(tm, MI_FRESNEL_DISNEY) && !TM_HAS(tm, MI_BXDF_FRESNEL_BLEND) && \
                             !TM_HAS(tm, MI_BXDF_DISNEY_CLEARCOAT) && !TM_HAS(tm, MI_BXDF_DISNEY_DIFFUSE) && !TM_HAS(tm, MI_BXDF_DISNEY_RETRO) && (((tm) & (TM_TEXTURED | TM_SCALED)) == 0u))
struct DivTrack {
    unsigned mx, mn;   // over the quotient candidates q1 != 0 formed so far: max of bits(|q1|), min of bits(|q1|) - 1
    DEV void Reset(bool allDivisorsFast) { mx = allDivisorsFast ? 0u : 0xffffffffu; mn = 0xffffffffu; }
    // DivBy's `aq > 1e-30f && aq < 1e30f` for all of them (bits of 1e-30f: 0x0da24260, of 1e30f: 0x7149f2ca)
    DEV bool Bad() const { return mx >= 0x7149f2cau || mn < 0x0da24260u; }
};
DEV float DivFast(float x, float d, float r, DivTrack &t) {
    const float q = x * r;
    const float rem = __builtin_fmaf(-q, d, x);
    const float q1 = __builtin_fmaf(rem, r, q);
    const unsigned u = __float_as_uint(q1) & 0x7fffffffu;
    t.mx = max(t.mx, u);
    t.mn = min(t.mn, u - 1u);   // (u == 0: wraps to the top, leaves the minimum alone)
    return q1;
}
// The lobe list of one evaluation in canonical form, per lane: f[bin] = sum over slots of ((R*A)*B)*C / d. A slot the lane does
// not use has A = 0 (its term is +0: f + 0 == f), a slot without division d = r = 1 (DivBy(v, 1) == v). TM says at compile
// time which factors can occur at all (matte lobes have neither c nor d).
#define TM_HAS_DIV_KINDS(tm) (TM_HAS(tm, MI_BXDF_MICROFACET_REFLECTION) || TM_SPECULAR(tm))
#define TM_HAS_C_FACTOR(tm) (TM_HAS(tm, MI_BXDF_MICROFACET_REFLECTION))
template <int NL>
struct SimpleLobes {
    float A[NL], B[NL], C[NL], d[NL], r[NL];
    unsigned off[NL];                              // byte offset of the slot's spectrum (R, or S for bit 8 of LobeEval::lobe) from bx
    bool anyOn[NL], anyB[NL], anyC[NL], anyDiv[NL];   // wave-uniform: some lane needs the slot / the factor / the division
    bool divisorsFast;                             // per lane: every divisor of the list is in DivBy's fast range
};
template <int NL, unsigned TM>
DEV SimpleLobes<NL> MakeSimpleLobes(const BSDFEvalT<NL> &ev) {
    SimpleLobes<NL> sl;
    sl.divisorsFast = true;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const LobeEval &le = ev.lobes[i];
        const bool on = i < ev.n;
        const int kind = on ? (le.kind & 0xff) : LK_NONE;
        const bool hasB = kind == LK_MUL2 || kind == LK_MUL3 || kind == LK_MUL3_DIV || kind == LK_MUL2_DIV || kind == LK_MTRANS;
        const bool hasC = TM_HAS_C_FACTOR(TM) && (kind == LK_MUL3 || kind == LK_MUL3_DIV);
        const bool div = TM_HAS_DIV_KINDS(TM) && (kind == LK_MUL3_DIV || kind == LK_MUL1_DIV || kind == LK_MUL2_DIV);
        sl.A[i] = !on ? 0.f : ((kind == LK_MTRANS) ? (1.f - le.a) : le.a);
        sl.B[i] = hasB ? le.b : 1.f;
        sl.C[i] = hasC ? le.c : 1.f;
        sl.d[i] = div ? le.d : 1.f;
        sl.r[i] = div ? le.r : 1.f;
        if (div && !(le.kind & LK_FASTDIV)) sl.divisorsFast = false;
        const int li = on ? (le.lobe & 0xff) : 0;
        sl.off[i] = (unsigned)li * (unsigned)sizeof(mi_bxdf) + ((on && (le.lobe & 0x100)) ? (unsigned)offsetof(mi_bxdf, S) : (unsigned)offsetof(mi_bxdf, R));
        sl.anyOn[i] = __any(on);
        sl.anyB[i] = __any(hasB); sl.anyC[i] = __any(hasC); sl.anyDiv[i] = __any(div);
    }
    return sl;
}
DEV Divisor DivisorOf(float d, float r) {
    Divisor v;
    v.d = d; v.r = r;
    const float a = absf(d);
    v.fast = (a > 1e-18f) && (a < 1e18f);
    return v;
}
// Quad c of f = sum of the lobes (EvalQuad's value). EXACT: the defining form, every quotient by DivBy (taken when the
// fast form's range test fails somewhere in the quad).
template <int NL, unsigned TM, bool EXACT>
DEV float4 SimpleEvalQuad(const SimpleLobes<NL> &sl, const mi_bxdf *bx, int c, DivTrack &t) {
    float f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        if (!sl.anyOn[i]) continue;   // (wave-uniform)
        const float4 R = LoadSpec4(reinterpret_cast<const float *>(reinterpret_cast<const char *>(bx) + sl.off[i]), c);
        float v[4] = {R.x * sl.A[i], R.y * sl.A[i], R.z * sl.A[i], R.w * sl.A[i]};
        if (sl.anyB[i]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] *= sl.B[i];
        }
        if (TM_HAS_C_FACTOR(TM) && sl.anyC[i]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] *= sl.C[i];
        }
        if (TM_HAS_DIV_KINDS(TM) && sl.anyDiv[i]) {
            if constexpr (EXACT) {
                const Divisor dv = DivisorOf(sl.d[i], sl.r[i]);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (sl.d[i] == 1.f && sl.r[i] == 1.f) ? v[k] : DivBy(v[k], dv);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = DivFast(v[k], sl.d[i], sl.r[i], t);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] += v[k];
    }
    return make_float4(f[0], f[1], f[2], f[3]);
}
// `acc |= bits(|x|)`: acc != 0 afterwards iff some x was != 0 (a NaN counts, as in `x != 0.f`)
DEV void OrNonZero(unsigned &acc, float x) { acc |= __float_as_uint(x) & 0x7fffffffu; }

// BSDF::f(woW, wiW, flags) (reflection.cpp:670-683) summed lobe by lobe into the caller's spectrum (rd(c) / wr(c, quad): quad c of
// the lane's column of the LDS tile), for the instances with more than two lobes: ONE lobe description alive at a time and
// one copy of LobeF's switch, where the list form keeps NL descriptions (73 registers at NL = 8, indexed at run time: the
// Disney instance ran them through 2 500 scratch instructions) behind NL unrolled copies of the switch (96 k instructions),
// and EvalBin fetched every bin of every lobe with a load of its own. The sum runs in lobe order, from 0.f: the same
// additions as EvalBin's. Returns the number of lobes that contributed.
template <int NL, unsigned TM, typename RD, typename WR>
DEV int AccumulateLobe(const LobeEval &le, const mi_bxdf *bx, const LobeTexT<NL> *lt, RD rd, WR wr) {
    const int li = le.lobe & 0xff;
    const mi_bxdf &b = bx[li];
#pragma unroll 1
    for (int c = 0; c < 8; ++c) {
        float4 acc = rd(c);
        float R[4], Sv[4] = {0.f, 0.f, 0.f, 0.f}, Kv[4] = {0.f, 0.f, 0.f, 0.f}, Sc[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr ((TM & TM_TEXTURED) != 0) {
            const float4 r4 = TexturedQuad(*lt, b, li, (le.lobe & 0x100) ? 1 : 0, c);
            R[0] = r4.x; R[1] = r4.y; R[2] = r4.z; R[3] = r4.w;
            if constexpr (TM_NEEDS_S(TM)) { const float4 s4 = TexturedQuad(*lt, b, li, 1, c); Sv[0] = s4.x; Sv[1] = s4.y; Sv[2] = s4.z; Sv[3] = s4.w; }
        } else {
            const float4 r4 = LoadSpec4((le.lobe & 0x100) ? b.S : b.R, c);
            R[0] = r4.x; R[1] = r4.y; R[2] = r4.z; R[3] = r4.w;
            if constexpr (TM_NEEDS_S(TM)) { const float4 s4 = LoadSpec4(b.S, c); Sv[0] = s4.x; Sv[1] = s4.y; Sv[2] = s4.z; Sv[3] = s4.w; }
        }
        if constexpr (TM_NEEDS_K(TM)) {
            float4 k4 = LoadSpec4(b.K, c);
            if constexpr ((TM & TM_TEXTURED) != 0) {   // "metal" with an image-textured k: its texture sits in the lobe's R slot
                if ((lt->hasK >> li) & 1u) k4 = TexQuad(lt->basis, lt->textures, lt->r[li], c);
            }
            Kv[0] = k4.x; Kv[1] = k4.y; Kv[2] = k4.z; Kv[3] = k4.w;
        }
        const bool scaled = (TM & TM_SCALED) != 0 && (le.lobe & 0x200);
        const bool scaled2 = scaled && (le.lobe & 0x400);
        float Sc2[4] = {1.f, 1.f, 1.f, 1.f};
        if (scaled) { const float4 c4 = LoadSpec4(b.scale, c); Sc[0] = c4.x; Sc[1] = c4.y; Sc[2] = c4.z; Sc[3] = c4.w; }
        if (scaled2) { const float4 c4 = LoadSpec4(b.scale2, c); Sc2[0] = c4.x; Sc2[1] = c4.y; Sc2[2] = c4.z; Sc2[3] = c4.w; }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = LobeValueCore<TM>(le, R[k], Sv[k], Kv[k]);
            if (scaled) v[k] = Sc[k] * v[k];
            if (scaled2) v[k] = Sc2[k] * v[k];
        }
        if (c == 7) v[3] = 0.f;   // bin 31 does not exist
        acc.x += v[0]; acc.y += v[1]; acc.z += v[2]; acc.w += v[3];
        wr(c, acc);
    }
    return 1;
}
// (wo, wi: in the shading frame; reflect: Dot(wiW, ng) * Dot(woW, ng) > 0)
template <int NL, unsigned TM, typename RD, typename WR>
DEV int AccumulateFLocal(const BSDFFrame &fr, const V3 &wo, const V3 &wi, bool reflect, int flags, const LobeTexT<NL> *lt, RD rd, WR wr) {
#pragma unroll 1
    for (int c = 0; c < 8; ++c) wr(c, make_float4(0.f, 0.f, 0.f, 0.f));
    if (wo.z == 0) return 0;
    const mi_material *m = fr.m;
    int n = 0;
#pragma unroll 1
    for (int i = 0; i < m->n_bxdfs; ++i) {
        if (!fr.On(i)) continue;
        const mi_bxdf &b = m->bxdf[i];
        if (!(MatchesFlags(b, flags) && ((reflect && (b.flags & MI_BSDF_REFLECTION)) || (!reflect && (b.flags & MI_BSDF_TRANSMISSION))))) continue;
        const LobeEval le = LobeF<TM>(b, i, wo, wi, fr.ov);
        if ((le.kind & 0xff) == LK_NONE) continue;
        n += AccumulateLobe<NL, TM>(le, m->bxdf, lt, rd, wr);
    }
    return n;
}
template <int NL, unsigned TM, typename RD, typename WR>
DEV int AccumulateF(const BSDFFrame &fr, const V3 &woW, const V3 &wiW, int flags, const LobeTexT<NL> *lt, RD rd, WR wr) {
    return AccumulateFLocal<NL, TM>(fr, fr.WorldToLocal(woW), fr.WorldToLocal(wiW), Dot(wiW, fr.ng) * Dot(woW, fr.ng) > 0, flags, lt, rd, wr);
}
// (a specular sample's single lobe, the same way)
template <int NL, unsigned TM, typename RD, typename WR>
DEV void AccumulateSpecular(const LobeEval &le, const mi_bxdf *bx, const LobeTexT<NL> *lt, RD rd, WR wr) {
#pragma unroll 1
    for (int c = 0; c < 8; ++c) wr(c, make_float4(0.f, 0.f, 0.f, 0.f));
    AccumulateLobe<NL, TM>(le, bx, lt, rd, wr);
}

// BSDF::Sample_f (reflection.cpp:703-768). Returns false when the reference returns a
// black f (including the early-outs that leave *pdf untouched). On success the value is
// described by *ev (one specular LobeEval, or the lobe list for the sampled direction).
// FILL = false (the shading instances with more than two lobes): a non-specular sample leaves the lobe list empty -- the
// caller sums f over the lobes straight into its spectrum tile (AccumulateF) -- a specular one still describes its single lobe.
template <int NL, unsigned TM, bool FILL = true>
DEV bool BSDF_Sample_f(const BSDFFrame &fr, const V3 &woWorld, V3 *wiWorld, float u0, float u1, float *pdf, int type,
                       int *sampledType, BSDFEvalT<NL> *ev, V3 *wiLocalOut = nullptr) {
    const mi_material *m = fr.m;
    const AlphaOv &ov = fr.ov;
    ev->n = 0;
    int matchingComps = NumComponents(fr, type);
    if (matchingComps == 0) { *pdf = 0; *sampledType = 0; return false; }
    int comp = min((int)floorf(u0 * matchingComps), matchingComps - 1);
    int bi = -1, count = comp;
    for (int i = 0; i < m->n_bxdfs; ++i)
        if (fr.On(i) && MatchesFlags(m->bxdf[i], type) && count-- == 0) { bi = i; break; }
    const mi_bxdf &b = m->bxdf[bi];
    float ur0 = minf(u0 * matchingComps - comp, kOneMinusEpsilon), ur1 = u1;
    V3 wi, wo = fr.WorldToLocal(woWorld);
    if (wo.z == 0) return false;
    *pdf = 0;
    *sampledType = b.flags;
    LobeEval spec;
    spec.kind = LK_NONE; spec.lobe = bi | (b.scaled ? 0x200 : 0) | (b.scaled >= 2 ? 0x400 : 0); spec.a = spec.b = spec.c = spec.d = spec.e = spec.f = spec.r = 0;
    bool isSpecular = (b.flags & MI_BSDF_SPECULAR) != 0;
    switch (b.type) {
    case MI_BXDF_SPECULAR_REFLECTION: if constexpr (TM_HAS(TM, MI_BXDF_SPECULAR_REFLECTION)) {  // (F*R)/|cos|
        wi = V3(-wo.x, -wo.y, wo.z);
        *pdf = 1;
        float F = (b.fresnel == MI_FRESNEL_DIELECTRIC) ? FrDielectric(CosTheta(wi), b.p[2], b.p[3]) : 1.f;
        spec.kind = LK_MUL1_DIV; spec.a = F; SetDivisor(spec, AbsCosTheta(wi));
        break;
    } break;
    case MI_BXDF_SPECULAR_TRANSMISSION: if constexpr (TM_HAS(TM, MI_BXDF_SPECULAR_TRANSMISSION)) {  // ((T*(1-F))*ratio)/|cos|
        const float etaA = b.p[0], etaB = b.p[1];
        bool entering = CosTheta(wo) > 0;
        float etaI = entering ? etaA : etaB, etaT = entering ? etaB : etaA;
        if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, &wi)) break;
        *pdf = 1;
        spec.kind = LK_MUL2_DIV;
        spec.a = (1.f - FrDielectric(CosTheta(wi), etaA, etaB));
        spec.b = (etaI * etaI) / (etaT * etaT);
        SetDivisor(spec, AbsCosTheta(wi));
        break;
    } break;
    case MI_BXDF_FRESNEL_SPECULAR: if constexpr (TM_HAS(TM, MI_BXDF_FRESNEL_SPECULAR)) {
        const float etaA = b.p[0], etaB = b.p[1];
        float F = FrDielectric(CosTheta(wo), etaA, etaB);
        if (ur0 < F) {
            wi = V3(-wo.x, -wo.y, wo.z);
            *sampledType = MI_BSDF_SPECULAR | MI_BSDF_REFLECTION;
            *pdf = F;
            spec.kind = LK_MUL1_DIV; spec.a = F; SetDivisor(spec, AbsCosTheta(wi));
        } else {
            bool entering = CosTheta(wo) > 0;
            float etaI = entering ? etaA : etaB, etaT = entering ? etaB : etaA;
            if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, &wi)) break;
            *sampledType = MI_BSDF_SPECULAR | MI_BSDF_TRANSMISSION;
            *pdf = 1 - F;
            spec.kind = LK_MUL2_DIV; spec.lobe |= 0x100;  // bit 8: use S (= T) instead of R
            spec.a = (1 - F); spec.b = (etaI * etaI) / (etaT * etaT); SetDivisor(spec, AbsCosTheta(wi));
        }
        break;
    } break;
    case MI_BXDF_FRESNEL_BLEND: if constexpr (TM_HAS(TM, MI_BXDF_FRESNEL_BLEND)) {  // reflection.cpp:450-468
        if ((double)ur0 < .5) {
            const float v0 = minf(2 * ur0, kOneMinusEpsilon);
            wi = CosineSampleHemisphere(v0, ur1);
            if (wo.z < 0) wi.z *= -1;
        } else {
            const float v0 = minf(2 * (ur0 - .5f), kOneMinusEpsilon);
            V3 wh = DistOf<TM>(b, ov).Sample_wh(wo, v0, ur1);
            wi = Reflect(wo, wh);
            if (!SameHemisphere(wo, wi)) break;
        }
        *pdf = LobePdf<TM>(b, wo, wi, fr.ov);
        break;
    } break;
    case MI_BXDF_LAMBERTIAN_TRANSMISSION: if constexpr (TM_HAS(TM, MI_BXDF_LAMBERTIAN_TRANSMISSION)) {
        wi = CosineSampleHemisphere(ur0, ur1);
        if (wo.z > 0) wi.z *= -1;
        *pdf = LobePdf<TM>(b, wo, wi, fr.ov);
        break;
    } break;
    case MI_BXDF_MICROFACET_REFLECTION: if constexpr (TM_HAS(TM, MI_BXDF_MICROFACET_REFLECTION)) {
        V3 wh = DistOf<TM>(b, ov).Sample_wh(wo, ur0, ur1);
        wi = Reflect(wo, wh);
        if (!SameHemisphere(wo, wi)) break;
        *pdf = DistOf<TM>(b, ov).Pdf(wo, wh) / (4 * Dot(wo, wh));
        break;
    } break;
    case MI_BXDF_MICROFACET_TRANSMISSION: if constexpr (TM_HAS(TM, MI_BXDF_MICROFACET_TRANSMISSION)) {
        V3 wh = DistOf<TM>(b, ov).Sample_wh(wo, ur0, ur1);
        float eta = CosTheta(wo) > 0 ? (b.p[2] / b.p[3]) : (b.p[3] / b.p[2]);
        if (!Refract(wo, wh, eta, &wi)) break;
        *pdf = LobePdf<TM>(b, wo, wi, fr.ov);
        break;
    } break;
    case MI_BXDF_DISNEY_CLEARCOAT: if constexpr (TM_HAS(TM, MI_BXDF_DISNEY_CLEARCOAT)) {
        float alpha2 = b.p[1] * b.p[1];
        float cosTheta = __builtin_sqrtf(maxf(0.f, (1 - powF(alpha2, 1 - ur0)) / (1 - alpha2)));
        float sinTheta = __builtin_sqrtf(maxf(0.f, 1 - cosTheta * cosTheta));
        float phi = 2 * kPi * ur1;
        V3 wh = SphericalDirection(sinTheta, cosTheta, phi);
        if (!SameHemisphere(wo, wh)) wh = -wh;
        wi = Reflect(wo, wh);
        if (!SameHemisphere(wo, wi)) break;
        *pdf = LobePdf<TM>(b, wo, wi, fr.ov);
        break;
    } break;
    default: {
        wi = CosineSampleHemisphere(ur0, ur1);
        if (wo.z < 0) wi.z *= -1;
        *pdf = LobePdf<TM>(b, wo, wi, fr.ov);
        break;
    }
    }
    if (*pdf == 0) { *sampledType = 0; return false; }
    *wiWorld = fr.LocalToWorld(wi);
    if (wiLocalOut) *wiLocalOut = wi;   // (the lobes are evaluated with the sampled local direction itself, not with its way back from the world)
    if (!isSpecular && matchingComps > 1)
        for (int i = 0; i < m->n_bxdfs; ++i)
            if (i != bi && fr.On(i) && MatchesFlags(m->bxdf[i], type)) *pdf += LobePdf<TM>(m->bxdf[i], wo, wi, fr.ov);
    if (matchingComps > 1) *pdf /= matchingComps;
    if (isSpecular) {
        ev->n = 1;
        ev->lobes[0] = spec;
    } else if constexpr (FILL) {
        bool reflect = Dot(*wiWorld, fr.ng) * Dot(woWorld, fr.ng) > 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (i < m->n_bxdfs && fr.On(i)) {
                const mi_bxdf &bb = m->bxdf[i];
                if (MatchesFlags(bb, type) &&
                    ((reflect && (bb.flags & MI_BSDF_REFLECTION)) || (!reflect && (bb.flags & MI_BSDF_TRANSMISSION)))) {
                    LobeEval le = LobeF<TM>(bb, i, wo, wi, fr.ov);
                    if ((le.kind & 0xff) != LK_NONE) ev->lobes[ev->n++] = le;
                }
            }
        }
    }
    return true;
}

}  // namespace dpt
