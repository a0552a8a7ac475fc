// oracle/o_bsdf.h -- TEST INFRASTRUCTURE (CPU oracle).
// BSDF / BxDF evaluation over the compiled lobe list of include/mi_pt.h, restating
// src/core/reflection.{h,cpp} (BSDF::f 670-683, Sample_f 703-768, Pdf 770-785, BxDFs
// 47-511), src/core/microfacet.{h,cpp} (Trowbridge-Reitz) and the Disney lobes of
// src/materials/disney.cpp:72-356.
#pragma once
#include "../include/mi_pt.h"
#include "o_math.h"
#include "o_shapes.h"
#include "o_texture.h"

namespace orc {

inline Float SpecY(const mi_scene_desc &d, const Spec &s) {  // SampledSpectrum::y(), spectrum.h:415-421
    Float yy = 0.f;
    for (int i = 0; i < NS; ++i) yy += d.cie_y[i] * s.c[i];
    yy = (yy < 0) ? 0 : yy;
    return yy * Float(705 - 395) / Float(106.856895f * NS);
}


// ---- local-frame trig, reflection.h:50-89
inline Float CosTheta(const V3 &w) { return w.z; }
inline Float Cos2Theta(const V3 &w) { return w.z * w.z; }
inline Float AbsCosTheta(const V3 &w) { return std::abs(w.z); }
inline Float Sin2Theta(const V3 &w) { return std::max((Float)0, (Float)1 - Cos2Theta(w)); }
inline Float SinTheta(const V3 &w) { return std::sqrt(Sin2Theta(w)); }
inline Float TanTheta(const V3 &w) { return SinTheta(w) / CosTheta(w); }
inline Float Tan2Theta(const V3 &w) { return Sin2Theta(w) / Cos2Theta(w); }
inline Float CosPhi(const V3 &w) { Float s = SinTheta(w); return (s == 0) ? 1 : Clamp(w.x / s, -1, 1); }
inline Float SinPhi(const V3 &w) { Float s = SinTheta(w); return (s == 0) ? 0 : Clamp(w.y / s, -1, 1); }
inline Float Cos2Phi(const V3 &w) { return CosPhi(w) * CosPhi(w); }
inline Float Sin2Phi(const V3 &w) { return SinPhi(w) * SinPhi(w); }
inline V3 Reflect(const V3 &wo, const V3 &n) { return -wo + 2 * Dot(wo, n) * n; }
inline bool Refract(const V3 &wi, const V3 &n, Float eta, V3 *wt) {  // reflection.h:95-108
    Float cosThetaI = Dot(n, wi);
    Float sin2ThetaI = std::max(Float(0), Float(1 - cosThetaI * cosThetaI));
    Float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    Float cosThetaT = std::sqrt(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}
inline bool SameHemisphere(const V3 &w, const V3 &wp) { return w.z * wp.z > 0; }

inline Float FrDielectric(Float cosThetaI, Float etaI, Float etaT) {  // reflection.cpp:47-69
    cosThetaI = Clamp(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) { std::swap(etaI, etaT); cosThetaI = std::abs(cosThetaI); }
    Float sinThetaI = std::sqrt(std::max((Float)0, 1 - cosThetaI * cosThetaI));
    Float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    Float cosThetaT = std::sqrt(std::max((Float)0, 1 - sinThetaT * sinThetaT));
    Float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    Float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}

inline void ConcentricSampleDisk(const Float u[2], Float d[2]) {  // sampling.cpp:113-130
    Float ox = 2.f * u[0] - 1, oy = 2.f * u[1] - 1;
    if (ox == 0 && oy == 0) { d[0] = 0; d[1] = 0; return; }
    Float theta, r;
    if (std::abs(ox) > std::abs(oy)) { r = ox; theta = PiOver4 * (oy / ox); }
    else { r = oy; theta = PiOver2 - PiOver4 * (ox / oy); }
    d[0] = r * CosF(theta);
    d[1] = r * SinF(theta);
}
inline V3 CosineSampleHemisphere(const Float u[2]) {  // sampling.h:159-163
    Float d[2];
    ConcentricSampleDisk(u, d);
    Float z = std::sqrt(std::max((Float)0, 1 - d[0] * d[0] - d[1] * d[1]));
    return V3(d[0], d[1], z);
}

// ---- Trowbridge-Reitz, microfacet.cpp:165-184,238-336
struct TRDist {
    Float alphax, alphay;
    bool separableG;  // DisneyMicrofacetDistribution::G, disney.cpp:350-354
    Float D(const V3 &wh) const {
        Float tan2Theta = Tan2Theta(wh);
        if (std::isinf(tan2Theta)) return 0.;
        const Float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
        Float e = (Cos2Phi(wh) / (alphax * alphax) + Sin2Phi(wh) / (alphay * alphay)) * tan2Theta;
        return 1 / (Pi * alphax * alphay * cos4Theta * (1 + e) * (1 + e));
    }
    Float Lambda(const V3 &w) const {
        Float absTanTheta = std::abs(TanTheta(w));
        if (std::isinf(absTanTheta)) return 0.;
        Float alpha = std::sqrt(Cos2Phi(w) * alphax * alphax + Sin2Phi(w) * alphay * alphay);
        Float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
        return (-1 + std::sqrt(1.f + alpha2Tan2Theta)) / 2;
    }
    Float G1(const V3 &w) const { return 1 / (1 + Lambda(w)); }
    Float G(const V3 &wo, const V3 &wi) const {
        if (separableG) return G1(wo) * G1(wi);
        return 1 / (1 + Lambda(wo) + Lambda(wi));
    }
    Float Pdf(const V3 &wo, const V3 &wh) const {  // sampleVisibleArea = true
        return D(wh) * G1(wo) * AbsDot(wo, wh) / AbsCosTheta(wo);
    }
    static void Sample11(Float cosTheta, Float U1, Float U2, Float *slope_x, Float *slope_y) {
        if (cosTheta > .9999) {
            // unqualified sqrt/cos/sin on Float arguments resolve to the double C
            // functions in the reference's translation unit; double results rounded
            // to float equal the correctly rounded float results.
            Float r = (Float)std::sqrt((double)(U1 / (1 - U1)));
            Float phi = 6.28318530718 * U2;
            *slope_x = r * (Float)std::cos((double)phi);
            *slope_y = r * (Float)std::sin((double)phi);
            return;
        }
        Float sinTheta = std::sqrt(std::max((Float)0, (Float)1 - cosTheta * cosTheta));
        Float tanTheta = sinTheta / cosTheta;
        Float a = 1 / tanTheta;
        Float G1 = 2 / (1 + std::sqrt(1.f + 1.f / (a * a)));
        Float A = 2 * U1 / G1 - 1;
        Float tmp = 1.f / (A * A - 1.f);
        if (tmp > 1e10) tmp = 1e10;
        Float B = tanTheta;
        Float D = std::sqrt(std::max(Float(B * B * tmp * tmp - (A * A - B * B) * tmp), Float(0)));
        Float slope_x_1 = B * tmp - D;
        Float slope_x_2 = B * tmp + D;
        *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
        Float S;
        if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
        else { S = -1.f; U2 = 2.f * (.5f - U2); }
        Float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        *slope_y = S * z * std::sqrt(1.f + *slope_x * *slope_x);
    }
    V3 Sample_wh(const V3 &wo, const Float u[2]) const {
        bool flip = wo.z < 0;
        V3 wi = flip ? -wo : wo;
        V3 wiStretched = Normalize(V3(alphax * wi.x, alphay * wi.y, wi.z));
        Float slope_x, slope_y;
        Sample11(CosTheta(wiStretched), u[0], u[1], &slope_x, &slope_y);
        Float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
        slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = alphax * slope_x;
        slope_y = alphay * slope_y;
        V3 wh = Normalize(V3(-slope_x, -slope_y, 1.));
        if (flip) wh = -wh;
        return wh;
    }
};

// ---- Disney helpers, disney.cpp:61-87,243-255
inline Float sqr(Float x) { return x * x; }
inline Float SchlickWeight(Float cosTheta) { Float m = Clamp(1 - cosTheta, 0, 1); return (m * m) * (m * m) * m; }
inline Float FrSchlick(Float R0, Float cosTheta) { return Lerp(SchlickWeight(cosTheta), R0, 1); }
inline Spec FrSchlick(const Spec &R0, Float cosTheta) { return Lerp(SchlickWeight(cosTheta), R0, Spec(1.)); }
inline Float GTR1(Float cosTheta, Float alpha) {
    Float alpha2 = alpha * alpha;
    return (alpha2 - 1) / (Pi * LogF(alpha2) * (1 + (alpha2 - 1) * cosTheta * cosTheta));
}
inline Float smithG_GGX(Float cosTheta, Float alpha) {
    Float alpha2 = alpha * alpha;
    Float cosTheta2 = cosTheta * cosTheta;
    // unqualified sqrt -> double sqrt, rounded to float by the division below
    return 1 / (cosTheta + (Float)std::sqrt((double)(alpha2 + cosTheta2 - alpha2 * cosTheta2)));
}

// reflection.cpp:71-94
inline Spec FrConductor(Float cosThetaI, const Spec &etai, const Spec &etat, const Spec &k) {
    cosThetaI = Clamp(cosThetaI, -1, 1);
    Spec eta = etat / etai;
    Spec etak = k / etai;
    Float cosThetaI2 = cosThetaI * cosThetaI;
    Float sinThetaI2 = 1. - cosThetaI2;
    Spec eta2 = eta * eta;
    Spec etak2 = etak * etak;
    Spec t0 = eta2 - etak2 - Spec(sinThetaI2);
    Spec a2plusb2 = Sqrt(t0 * t0 + 4 * eta2 * etak2);
    Spec t1 = a2plusb2 + Spec(cosThetaI2);
    Spec a = Sqrt(0.5f * (a2plusb2 + t0));
    Spec t2 = (Float)2 * cosThetaI * a;
    Spec Rs = (t1 - t2) / (t1 + t2);
    Spec t3 = cosThetaI2 * a2plusb2 + Spec(sinThetaI2 * sinThetaI2);
    Spec t4 = t2 * sinThetaI2;
    Spec Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5 * (Rp + Rs);
}

// TrowbridgeReitzDistribution::RoughnessToAlpha, microfacet.h:140-145
inline Float RoughnessToAlphaF(Float roughness) {
    roughness = std::max(roughness, (Float)1e-3);
    Float x = LogF(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

struct BxDF {
    const mi_bxdf *b;
    bool texR = false, texS = false, texK = false;   // the spectrum of this hit comes from an image texture (mi_lobe_tex)
    Spec Rtex, Stex, Ktex;
    Spec R() const { return texR ? Rtex : Spec::From(b->R); }
    Spec S() const { return texS ? Stex : Spec::From(b->S); }
    Spec K() const { return texK ? Ktex : Spec::From(b->K); }
    Spec Scale() const { return Spec::From(b->scale); }
    bool MatchesFlags(int t) const { return (b->flags & t) == b->flags; }
    // roughness from float textures (mi_material.rough_tex): the alphas of this hit in place of b->p[0] / p[1]
    bool ovU = false, ovV = false;
    bool roughOv = false;     // "disney" with a roughness map (MI_ROUGH_DISNEY): the roughness of this hit in place of p[0] (FakeSS, Retro)
    Float roughHit = 0;
    int sigMode = 0;          // "matte" with a sigma map: 1 = OrenNayar with sigA / sigB, 2 = LambertianReflection (sig == 0)
    Float sigA = 0, sigB = 0;
    Float alphaU = 0, alphaV = 0;
    TRDist Dist() const { return TRDist{ovU ? alphaU : b->p[0], ovV ? alphaV : b->p[1], b->p[5] != 0.f}; }

    Spec Fresnel(Float cosI) const {
        switch (b->fresnel) {
        case MI_FRESNEL_DIELECTRIC: return Spec(FrDielectric(cosI, b->p[2], b->p[3]));
        case MI_FRESNEL_CONDUCTOR:  // FresnelConductor::Evaluate, reflection.cpp:119-121 (etaI = 1, metal.cpp:75)
            return FrConductor(std::abs(cosI), Spec(1.), S(), K());
        case MI_FRESNEL_DISNEY:  // disney.cpp:324-343
            return Lerp(b->p[2], Spec(FrDielectric(cosI, 1, b->p[3])), FrSchlick(S(), cosI));
        default: return Spec(1.);
        }
    }

    // ScaledBxDF (mix material), reflection.cpp:96-111
    // (scaled == 2: a "mix" of a "mix", ScaledBxDF(ScaledBxDF(lobe, scale), scale2))
    Spec Scaled(const Spec &v) const {
        if (!b->scaled) return v;
        const Spec v1 = Scale() * v;
        return b->scaled >= 2 ? Spec::From(b->scale2) * v1 : v1;
    }
    Spec f(const V3 &wo, const V3 &wi) const { return Scaled(fInner(wo, wi)); }
    Spec Sample_f(const V3 &wo, V3 *wi, const Float u[2], Float *pdf, int *sampledType) const {
        return Scaled(Sample_fInner(wo, wi, u, pdf, sampledType));
    }

    Spec fInner(const V3 &wo, const V3 &wi) const {
        switch (b->type) {
        case MI_BXDF_FRESNEL_BLEND: {  // reflection.cpp:285-298, reflection.h:485-488
            auto pow5 = [](Float v) { return (v * v) * (v * v) * v; };
            const Spec Rd = R(), Rs = S();
            Spec diffuse = (28.f / (23.f * Pi)) * Rd * (Spec(1.f) - Rs) * (1 - pow5(1 - .5f * AbsCosTheta(wi))) *
                           (1 - pow5(1 - .5f * AbsCosTheta(wo)));
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0);
            wh = Normalize(wh);
            Spec schlick = Rs + pow5(1 - Dot(wi, wh)) * (Spec(1.) - Rs);
            Spec specular = Dist().D(wh) / (4 * AbsDot(wi, wh) * std::max(AbsCosTheta(wi), AbsCosTheta(wo))) * schlick;
            return diffuse + specular;
        }
        case MI_BXDF_LAMBERTIAN_REFLECTION: return R() * InvPi;
        case MI_BXDF_LAMBERTIAN_TRANSMISSION: return R() * InvPi;
        case MI_BXDF_OREN_NAYAR: {  // reflection.cpp:178-200
            if (sigMode == 2) return R() * InvPi;   // (matte.cpp:59-60: sig == 0 at this hit)
            Float sinThetaI = SinTheta(wi), sinThetaO = SinTheta(wo);
            Float maxCos = 0;
            if (sinThetaI > 1e-4 && sinThetaO > 1e-4) {
                Float sinPhiI = SinPhi(wi), cosPhiI = CosPhi(wi);
                Float sinPhiO = SinPhi(wo), cosPhiO = CosPhi(wo);
                Float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
                maxCos = std::max((Float)0, dCos);
            }
            Float sinAlpha, tanBeta;
            if (AbsCosTheta(wi) > AbsCosTheta(wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / AbsCosTheta(wi); }
            else { sinAlpha = sinThetaI; tanBeta = sinThetaO / AbsCosTheta(wo); }
            return R() * InvPi * ((sigMode ? sigA : b->p[0]) + (sigMode ? sigB : b->p[1]) * maxCos * sinAlpha * tanBeta);
        }
        case MI_BXDF_MICROFACET_REFLECTION: {  // reflection.cpp:207-217
            Float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
            V3 wh = wi + wo;
            if (cosThetaI == 0 || cosThetaO == 0) return Spec(0.);
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Spec F = Fresnel(Dot(wi, wh));
            TRDist d = Dist();
            return R() * d.D(wh) * d.G(wo, wi) * F / (4 * cosThetaI * cosThetaO);
        }
        case MI_BXDF_MICROFACET_TRANSMISSION: {  // reflection.cpp:226-249
            if (SameHemisphere(wo, wi)) return Spec(0);
            Float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
            if (cosThetaI == 0 || cosThetaO == 0) return Spec(0);
            const Float etaA = b->p[2], etaB = b->p[3];
            Float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
            V3 wh = Normalize(wo + wi * eta);
            if (wh.z < 0) wh = -wh;
            Spec F(FrDielectric(Dot(wo, wh), etaA, etaB));
            Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
            Float factor = 1 / eta;  // TransportMode::Radiance
            TRDist d = Dist();
            return (Spec(1.f) - F) * R() *
                   std::abs(d.D(wh) * d.G(wo, wi) * eta * eta * AbsDot(wi, wh) * AbsDot(wo, wh) * factor * factor /
                            (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
        }
        case MI_BXDF_DISNEY_DIFFUSE: {  // disney.cpp:104-111
            Float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
            return R() * InvPi * (1 - Fo / 2) * (1 - Fi / 2);
        }
        case MI_BXDF_DISNEY_FAKE_SS: {  // disney.cpp:139-156
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Float cosThetaD = Dot(wi, wh);
            Float Fss90 = cosThetaD * cosThetaD * (roughOv ? roughHit : b->p[0]);
            Float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
            Float Fss = Lerp(Fo, 1.0, Fss90) * Lerp(Fi, 1.0, Fss90);
            Float ss = 1.25f * (Fss * (1 / (AbsCosTheta(wo) + AbsCosTheta(wi)) - .5f) + .5f);
            return R() * InvPi * ss;
        }
        case MI_BXDF_DISNEY_RETRO: {  // disney.cpp:182-194
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Float cosThetaD = Dot(wi, wh);
            Float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
            Float Rr = 2 * (roughOv ? roughHit : b->p[0]) * cosThetaD * cosThetaD;
            return R() * InvPi * Rr * (Fo + Fi + Fo * Fi * (Rr - 1));
        }
        case MI_BXDF_DISNEY_SHEEN: {  // disney.cpp:217-224
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Float cosThetaD = Dot(wi, wh);
            return R() * SchlickWeight(cosThetaD);
        }
        case MI_BXDF_DISNEY_CLEARCOAT: {  // disney.cpp:263-278
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Float Dr = GTR1(AbsCosTheta(wh), b->p[1]);
            Float Fr = FrSchlick((Float).04, Dot(wo, wh));
            Float Gr = smithG_GGX(AbsCosTheta(wo), .25) * smithG_GGX(AbsCosTheta(wi), .25);
            return Spec(b->p[0] * Gr * Fr * Dr / 4);
        }
        default: return Spec(0.f);  // specular lobes, reflection.h:310-312,335-337,362-364
        }
    }

    Float Pdf(const V3 &wo, const V3 &wi) const {
        switch (b->type) {
        case MI_BXDF_FRESNEL_BLEND: {  // reflection.cpp:470-475
            if (!SameHemisphere(wo, wi)) return 0;
            V3 wh = Normalize(wo + wi);
            Float pdf_wh = Dist().Pdf(wo, wh);
            return .5f * (AbsCosTheta(wi) * InvPi + pdf_wh / (4 * Dot(wo, wh)));
        }
        case MI_BXDF_SPECULAR_REFLECTION: case MI_BXDF_SPECULAR_TRANSMISSION: case MI_BXDF_FRESNEL_SPECULAR:
            return 0;
        case MI_BXDF_LAMBERTIAN_TRANSMISSION: return !SameHemisphere(wo, wi) ? AbsCosTheta(wi) * InvPi : 0;
        case MI_BXDF_MICROFACET_REFLECTION: {  // reflection.cpp:414-418
            if (!SameHemisphere(wo, wi)) return 0;
            V3 wh = Normalize(wo + wi);
            return Dist().Pdf(wo, wh) / (4 * Dot(wo, wh));
        }
        case MI_BXDF_MICROFACET_TRANSMISSION: {  // reflection.cpp:432-445
            if (SameHemisphere(wo, wi)) return 0;
            const Float etaA = b->p[2], etaB = b->p[3];
            Float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
            V3 wh = Normalize(wo + wi * eta);
            Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
            Float dwh_dwi = std::abs((eta * eta * Dot(wi, wh)) / (sqrtDenom * sqrtDenom));
            return Dist().Pdf(wo, wh) * dwh_dwi;
        }
        case MI_BXDF_DISNEY_CLEARCOAT: {  // disney.cpp:305-319
            if (!SameHemisphere(wo, wi)) return 0;
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return 0;
            wh = Normalize(wh);
            Float Dr = GTR1(AbsCosTheta(wh), b->p[1]);
            return Dr * AbsCosTheta(wh) / (4 * Dot(wo, wh));
        }
        default:  // BxDF::Pdf, reflection.cpp:381-383
            return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * InvPi : 0;
        }
    }

    // Returns f; *pdf stays untouched when the reference leaves it untouched.
    Spec Sample_fInner(const V3 &wo, V3 *wi, const Float u[2], Float *pdf, int *sampledType) const {
        switch (b->type) {
        case MI_BXDF_FRESNEL_BLEND: {  // reflection.cpp:450-468
            Float uu[2] = {u[0], u[1]};
            if (uu[0] < .5) {
                uu[0] = std::min(2 * uu[0], OneMinusEpsilon);
                *wi = CosineSampleHemisphere(uu);
                if (wo.z < 0) wi->z *= -1;
            } else {
                uu[0] = std::min(2 * (uu[0] - .5f), OneMinusEpsilon);
                V3 wh = Dist().Sample_wh(wo, uu);
                *wi = Reflect(wo, wh);
                if (!SameHemisphere(wo, *wi)) return Spec(0.f);
            }
            *pdf = Pdf(wo, *wi);
            return fInner(wo, *wi);
        }
        case MI_BXDF_SPECULAR_REFLECTION: {  // reflection.cpp:127-134
            *wi = V3(-wo.x, -wo.y, wo.z);
            *pdf = 1;
            return Fresnel(CosTheta(*wi)) * R() / AbsCosTheta(*wi);
        }
        case MI_BXDF_SPECULAR_TRANSMISSION: {  // reflection.cpp:141-158
            const Float etaA = b->p[0], etaB = b->p[1];
            bool entering = CosTheta(wo) > 0;
            Float etaI = entering ? etaA : etaB;
            Float etaT = entering ? etaB : etaA;
            if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return Spec(0);
            *pdf = 1;
            Spec ft = R() * (Spec(1.) - Spec(FrDielectric(CosTheta(*wi), etaA, etaB)));
            ft *= (etaI * etaI) / (etaT * etaT);
            return ft / AbsCosTheta(*wi);
        }
        case MI_BXDF_FRESNEL_SPECULAR: {  // reflection.cpp:478-511
            const Float etaA = b->p[0], etaB = b->p[1];
            Float F = FrDielectric(CosTheta(wo), etaA, etaB);
            if (u[0] < F) {
                *wi = V3(-wo.x, -wo.y, wo.z);
                if (sampledType) *sampledType = MI_BSDF_SPECULAR | MI_BSDF_REFLECTION;
                *pdf = F;
                return F * R() / AbsCosTheta(*wi);
            } else {
                bool entering = CosTheta(wo) > 0;
                Float etaI = entering ? etaA : etaB;
                Float etaT = entering ? etaB : etaA;
                if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return Spec(0);
                Spec ft = S() * (1 - F);
                ft *= (etaI * etaI) / (etaT * etaT);
                if (sampledType) *sampledType = MI_BSDF_SPECULAR | MI_BSDF_TRANSMISSION;
                *pdf = 1 - F;
                return ft / AbsCosTheta(*wi);
            }
        }
        case MI_BXDF_LAMBERTIAN_TRANSMISSION: {  // reflection.cpp:385-392
            *wi = CosineSampleHemisphere(u);
            if (wo.z > 0) wi->z *= -1;
            *pdf = Pdf(wo, *wi);
            return fInner(wo, *wi);
        }
        case MI_BXDF_MICROFACET_REFLECTION: {  // reflection.cpp:399-412
            if (wo.z == 0) return Spec(0.);
            V3 wh = Dist().Sample_wh(wo, u);
            *wi = Reflect(wo, wh);
            if (!SameHemisphere(wo, *wi)) return Spec(0.f);
            *pdf = Dist().Pdf(wo, wh) / (4 * Dot(wo, wh));
            return fInner(wo, *wi);
        }
        case MI_BXDF_MICROFACET_TRANSMISSION: {  // reflection.cpp:420-430
            if (wo.z == 0) return Spec(0.);
            V3 wh = Dist().Sample_wh(wo, u);
            Float eta = CosTheta(wo) > 0 ? (b->p[2] / b->p[3]) : (b->p[3] / b->p[2]);
            if (!Refract(wo, wh, eta, wi)) return Spec(0);
            *pdf = Pdf(wo, *wi);
            return fInner(wo, *wi);
        }
        case MI_BXDF_DISNEY_CLEARCOAT: {  // disney.cpp:280-303
            if (wo.z == 0) return Spec(0.);
            Float alpha2 = b->p[1] * b->p[1];
            Float cosTheta = std::sqrt(std::max(Float(0), (1 - PowF(alpha2, 1 - u[0])) / (1 - alpha2)));
            Float sinTheta = std::sqrt(std::max((Float)0, 1 - cosTheta * cosTheta));
            Float phi = 2 * Pi * u[1];
            V3 wh = SphericalDirection(sinTheta, cosTheta, phi);
            if (!SameHemisphere(wo, wh)) wh = -wh;
            *wi = Reflect(wo, wh);
            if (!SameHemisphere(wo, *wi)) return Spec(0.f);
            *pdf = Pdf(wo, *wi);
            return fInner(wo, *wi);
        }
        default: {  // BxDF::Sample_f, reflection.cpp:371-379
            *wi = CosineSampleHemisphere(u);
            if (wo.z < 0) wi->z *= -1;
            *pdf = Pdf(wo, *wi);
            return fInner(wo, *wi);
        }
        }
    }
};

struct BSDF {
    Float eta;
    V3 ns, ng, ss, ts;
    int nBxDFs;
    BxDF bxdfs[MI_MAX_BXDFS];

    // d / td: scene and the hit's texture differentials, for materials with image-textured lobes
    // (Material::ComputeScatteringFunctions evaluates its textures at the hit and adds a lobe only when the
    // spectrum it tests is not black, e.g. matte.cpp:55-63, uber.cpp:60-100)
    BSDF(const SurfaceInteraction &si, const mi_material &m, const mi_scene_desc *d = nullptr, const TexDifferentials *td = nullptr) {  // reflection.h:170-176
        eta = m.eta;
        ns = si.shading.n;
        ng = si.n;
        ss = Normalize(si.shading.dpdu);
        ts = Cross(ns, ss);
        nBxDFs = 0;
        // `rough = roughness->Evaluate(*si); if (remapRoughness) rough = RoughnessToAlpha(rough)`, plastic.cpp:57-62 etc.
        bool ov[2] = {false, false};
        Float alpha[2] = {0, 0};
        Float raw[2] = {m.n_bxdfs > 0 ? m.bxdf[0].p[6] : 0, m.n_bxdfs > 0 ? m.bxdf[0].p[7] : 0};   // (MI_ROUGH_GLASS: the values before the remap)
        const bool disneyRough = d && td && (m.rough_flags & MI_ROUGH_DISNEY) != 0;
        const Float roughHit = disneyRough ? EvalFloatImageTexture(*d, m.rough_tex[0], si.uv[0], si.uv[1], *td) : 0;   // disney.cpp:491
        for (int a = 0; a < 2; ++a)
            if (d && td && m.rough_tex[a] >= 0 && !disneyRough) {
                Float r = (a == 1 && m.rough_tex[1] == m.rough_tex[0]) ? -1.f : EvalFloatImageTexture(*d, m.rough_tex[a], si.uv[0], si.uv[1], *td);
                if (a == 1 && m.rough_tex[1] == m.rough_tex[0]) { alpha[1] = alpha[0]; raw[1] = raw[0]; }
                else { alpha[a] = (m.rough_flags & MI_ROUGH_REMAP) ? RoughnessToAlphaF(r) : r; raw[a] = r; }
                ov[a] = true;
            }
        // glass.cpp:66: `bool isSpecular = urough == 0 && vrough == 0` at this hit
        const bool glassSwitch = d && td && (m.rough_flags & MI_ROUGH_GLASS) != 0;
        const bool glassSpecular = raw[0] == 0 && raw[1] == 0;
        // `Float sig = Clamp(sigma->Evaluate(*si), 0, 90)`, matte.cpp:57; OrenNayar's constructor, reflection.h:414-420
        int sigMode = 0;
        Float sigA = 0, sigB = 0;
        if (d && td && m.sigma_tex >= 0) {
            const Float sig = Clamp(EvalFloatImageTexture(*d, m.sigma_tex, si.uv[0], si.uv[1], *td), 0, 90);
            if (sig == 0) sigMode = 2;
            else {
                const Float sigma = (Pi / 180) * sig, sigma2 = sigma * sigma;   // Radians(), pbrt.h:252
                sigMode = 1;
                sigA = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                sigB = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
        }
        for (int i = 0; i < m.n_bxdfs; ++i) {
            BxDF bx;
            bx.b = &m.bxdf[i];
            bx.ovU = ov[0]; bx.ovV = ov[1]; bx.alphaU = alpha[0]; bx.alphaV = alpha[1];
            bx.sigMode = sigMode; bx.sigA = sigA; bx.sigB = sigB;
            if (disneyRough) {   // disney.cpp:538-541, 568-573
                const int t = bx.b->type;
                if (t == MI_BXDF_DISNEY_FAKE_SS || t == MI_BXDF_DISNEY_RETRO) { bx.roughOv = true; bx.roughHit = roughHit; }
                else if (t == MI_BXDF_MICROFACET_REFLECTION || t == MI_BXDF_MICROFACET_TRANSMISSION) {
                    const Float r = (t == MI_BXDF_MICROFACET_TRANSMISSION && bx.b->p[7] != 0) ? bx.b->p[7] * roughHit : roughHit;
                    bx.ovU = bx.ovV = true;
                    bx.alphaU = std::max(Float(.001), (r * r) / bx.b->p[4]);
                    bx.alphaV = std::max(Float(.001), (r * r) * bx.b->p[4]);
                }
            }
            if (glassSwitch && ((bx.b->type == MI_BXDF_FRESNEL_SPECULAR) != glassSpecular)) continue;
            const mi_lobe_tex &lt = m.tex[i];
            if (m.textured && d && td && lt.rule == MI_LOBE_METAL) {   // metal.cpp:119-122: eta and k from their textures, R = 1
                if (lt.tex_S >= 0) { bx.texS = true; bx.Stex = EvalImageTexture(*d, lt.tex_S, si, *td); }
                if (lt.tex_R >= 0) { bx.texK = true; bx.Ktex = EvalImageTexture(*d, lt.tex_R, si, *td); }
            } else
            if (m.textured && d && td && (lt.tex_R >= 0 || lt.tex_S >= 0)) {
                bool texBlack = true;
                if (lt.tex_R >= 0) {
                    const Spec T = EvalImageTexture(*d, lt.tex_R, si, *td);
                    texBlack = texBlack && T.IsBlack();
                    bx.texR = true;
                    bx.Rtex = (lt.flags & MI_LOBE_TEX_MUL_R) ? Spec::From(bx.b->R) * T : T;
                }
                if (lt.tex_S >= 0) {
                    const Spec T = EvalImageTexture(*d, lt.tex_S, si, *td);
                    texBlack = texBlack && T.IsBlack();
                    bx.texS = true;
                    bx.Stex = (lt.flags & MI_LOBE_TEX_MUL_S) ? Spec::From(bx.b->S) * T : T;
                }
                // "disney" with an image-textured colour (disney.cpp:485-587; mi_lobe_rule): the lobes are added whatever the
                // colour is, three of their spectra are not linear in it
                if (lt.rule >= MI_LOBE_DISNEY_SHEEN && lt.rule <= MI_LOBE_DISNEY_STRANS) {
                    const Spec c = EvalImageTexture(*d, lt.tex_R, si, *td);
                    const Float lum = SpecY(*d, c);
                    const Spec Ctint = lum > 0 ? (c / lum) : Spec(1.);
                    const Float *p = bx.b->p;
                    if (lt.rule == MI_LOBE_DISNEY_SHEEN) bx.Rtex = p[6] * Lerp(p[7], Spec(1.), Ctint);
                    else if (lt.rule == MI_LOBE_DISNEY_SPEC) { bx.Rtex = c; bx.texS = true; bx.Stex = Lerp(p[2], p[7] * Lerp(p[6], Spec(1.), Ctint), c); }
                    else bx.Rtex = p[6] * Sqrt(c);
                }
                bool present;
                if (lt.rule >= MI_LOBE_ALWAYS) present = true;
                else if (lt.rule == MI_LOBE_IF_R_OR_S) present = !bx.R().IsBlack() || !bx.S().IsBlack();
                else if (lt.rule == MI_LOBE_IF_TEX) present = !texBlack;
                else present = !bx.R().IsBlack();
                if (!present) continue;
            }
            bxdfs[nBxDFs++] = bx;
        }
    }
    int NumComponents(int flags) const {
        int num = 0;
        for (int i = 0; i < nBxDFs; ++i) if (bxdfs[i].MatchesFlags(flags)) ++num;
        return num;
    }
    V3 WorldToLocal(const V3 &v) const { return V3(Dot(v, ss), Dot(v, ts), Dot(v, ns)); }
    V3 LocalToWorld(const V3 &v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z,
                  ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    Spec f(const V3 &woW, const V3 &wiW, int flags) const {  // reflection.cpp:670-683
        V3 wi = WorldToLocal(wiW), wo = WorldToLocal(woW);
        if (wo.z == 0) return Spec(0.);
        bool reflect = Dot(wiW, ng) * Dot(woW, ng) > 0;
        Spec fv(0.f);
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(flags) &&
                ((reflect && (bxdfs[i].b->flags & MI_BSDF_REFLECTION)) ||
                 (!reflect && (bxdfs[i].b->flags & MI_BSDF_TRANSMISSION))))
                fv += bxdfs[i].f(wo, wi);
        return fv;
    }
    Float Pdf(const V3 &woW, const V3 &wiW, int flags) const {  // reflection.cpp:770-785
        if (nBxDFs == 0.f) return 0.f;
        V3 wo = WorldToLocal(woW), wi = WorldToLocal(wiW);
        if (wo.z == 0) return 0.;
        Float pdf = 0.f;
        int matchingComps = 0;
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(flags)) { ++matchingComps; pdf += bxdfs[i].Pdf(wo, wi); }
        return matchingComps > 0 ? pdf / matchingComps : 0.f;
    }
    // reflection.cpp:703-768. *pdf is written only where the reference writes it; the
    // callers initialise it to 0 (as EstimateDirect does; PathIntegrator::Li reads it
    // only after f.IsBlack() has short-circuited).
    Spec Sample_f(const V3 &woWorld, V3 *wiWorld, const Float u[2], Float *pdf, int type, int *sampledType) const {
        int matchingComps = NumComponents(type);
        if (matchingComps == 0) {
            *pdf = 0;
            if (sampledType) *sampledType = 0;
            return Spec(0);
        }
        int comp = std::min((int)std::floor(u[0] * matchingComps), matchingComps - 1);
        const BxDF *bxdf = nullptr;
        int count = comp;
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(type) && count-- == 0) { bxdf = &bxdfs[i]; break; }
        Float uRemapped[2] = {std::min(u[0] * matchingComps - comp, OneMinusEpsilon), u[1]};
        V3 wi, wo = WorldToLocal(woWorld);
        if (wo.z == 0) return Spec(0.);
        *pdf = 0;
        if (sampledType) *sampledType = bxdf->b->flags;
        Spec fv = bxdf->Sample_f(wo, &wi, uRemapped, pdf, sampledType);
        if (*pdf == 0) {
            if (sampledType) *sampledType = 0;
            return Spec(0);
        }
        *wiWorld = LocalToWorld(wi);
        if (!(bxdf->b->flags & MI_BSDF_SPECULAR) && matchingComps > 1)
            for (int i = 0; i < nBxDFs; ++i)
                if (&bxdfs[i] != bxdf && bxdfs[i].MatchesFlags(type)) *pdf += bxdfs[i].Pdf(wo, wi);
        if (matchingComps > 1) *pdf /= matchingComps;
        if (!(bxdf->b->flags & MI_BSDF_SPECULAR)) {
            bool reflect = Dot(*wiWorld, ng) * Dot(woWorld, ng) > 0;
            fv = Spec(0.);
            for (int i = 0; i < nBxDFs; ++i)
                if (bxdfs[i].MatchesFlags(type) &&
                    ((reflect && (bxdfs[i].b->flags & MI_BSDF_REFLECTION)) ||
                     (!reflect && (bxdfs[i].b->flags & MI_BSDF_TRANSMISSION))))
                    fv += bxdfs[i].f(wo, wi);
        }
        return fv;
    }
};

}  // namespace orc
