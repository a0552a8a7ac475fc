// materials.cpp -- compile a material with constant textures into the fixed BxDF
// list its ComputeScatteringFunctions() would build at every hit. Parameter names,
// defaults, clamping and lobe ORDER follow the reference (order matters: BSDF::Sample_f
// picks lobe floor(u*n), src/core/reflection.cpp:714-725).
#include <cmath>
#include "scene.h"

namespace mipt {
namespace {

#include "metal_copper_31.inc"

inline float RoughnessToAlpha(float roughness) {  // src/core/microfacet.h:140-145
    roughness = std::max(roughness, (float)1e-3);
    float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x +
           0.000640711f * x * x * x * x;
}
inline float sqr(float x) { return x * x; }

mi_bxdf MakeBxDF(int type, int flags, const Spectrum &R) {
    mi_bxdf b{};
    b.type = type;
    b.flags = flags;
    b.fresnel = MI_FRESNEL_NOOP;
    for (int i = 0; i < MI_NSPEC; ++i) b.R[i] = R.c[i];
    return b;
}
void SetS(mi_bxdf &b, const Spectrum &S) { for (int i = 0; i < MI_NSPEC; ++i) b.S[i] = S.c[i]; }

bool Add(mi_material *m, const mi_bxdf &b, std::vector<std::string> *errs) {
    if (m->n_bxdfs >= MI_MAX_BXDFS) { errs->push_back("more than 8 BxDFs in a BSDF"); return false; }
    m->tex[m->n_bxdfs] = mi_lobe_tex{-1, -1, 0u, MI_LOBE_IF_R};
    m->bxdf[m->n_bxdfs++] = b;
    return true;
}
// The lobe just added takes R (and S) from image textures at each hit (mi_lobe_tex, mi_pt.h).
void Bind(mi_material *m, int texR, bool mulR, int texS = -1, bool mulS = false, int rule = MI_LOBE_IF_R) {
    if (m->n_bxdfs == 0 || (texR < 0 && texS < 0)) return;
    mi_lobe_tex &t = m->tex[m->n_bxdfs - 1];
    t.tex_R = texR; t.tex_S = texS;
    t.flags = (texR >= 0 && mulR ? MI_LOBE_TEX_MUL_R : 0u) | (texS >= 0 && mulS ? MI_LOBE_TEX_MUL_S : 0u);
    t.rule = rule;
    m->textured = 1;
}
// (A parameter bound to `Texture "scale"` of an image texture and a constant c arrives as {s = c, tex, scaled}: the lobe
// keeps c -- clamped at 0, which for a texture value >= 0 is the clamp of the product -- and multiplies the texture
// value in at the hit. Where the material multiplies further (uber's opacity, translucent's reflect / transmit) the
// product is formed as (op * c) * T instead of the reference's op * (T * c): the last bit may differ.)
// Whether a parameter can make its lobe appear: a constant must not be black, a texture may be anything.
inline bool MayBeNonBlack(const SpectrumParam &p, const Spectrum &clamped) { return p.tex >= 0 || !clamped.IsBlack(); }

mi_bxdf Lambertian(const Spectrum &R) {
    return MakeBxDF(MI_BXDF_LAMBERTIAN_REFLECTION, MI_BSDF_REFLECTION | MI_BSDF_DIFFUSE, R);
}
mi_bxdf OrenNayar(const Spectrum &R, float sigma) {  // src/core/reflection.h:414-420
    mi_bxdf b = MakeBxDF(MI_BXDF_OREN_NAYAR, MI_BSDF_REFLECTION | MI_BSDF_DIFFUSE, R);
    sigma = Radians(sigma);
    float sigma2 = sigma * sigma;
    b.p[0] = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
    b.p[1] = 0.45f * sigma2 / (sigma2 + 0.09f);
    return b;
}
mi_bxdf MicrofacetReflectionDielectric(const Spectrum &R, float ax, float ay, float etaI, float etaT) {
    mi_bxdf b = MakeBxDF(MI_BXDF_MICROFACET_REFLECTION, MI_BSDF_REFLECTION | MI_BSDF_GLOSSY, R);
    b.fresnel = MI_FRESNEL_DIELECTRIC;
    b.p[0] = ax; b.p[1] = ay; b.p[2] = etaI; b.p[3] = etaT;
    return b;
}
mi_bxdf SpecularReflectionDielectric(const Spectrum &R, float etaI, float etaT) {
    mi_bxdf b = MakeBxDF(MI_BXDF_SPECULAR_REFLECTION, MI_BSDF_REFLECTION | MI_BSDF_SPECULAR, R);
    b.fresnel = MI_FRESNEL_DIELECTRIC;
    b.p[2] = etaI; b.p[3] = etaT;
    return b;
}
mi_bxdf SpecularTransmission(const Spectrum &T, float etaA, float etaB) {
    mi_bxdf b = MakeBxDF(MI_BXDF_SPECULAR_TRANSMISSION, MI_BSDF_TRANSMISSION | MI_BSDF_SPECULAR, T);
    b.p[0] = etaA; b.p[1] = etaB;
    return b;
}
mi_bxdf MicrofacetTransmission(const Spectrum &T, float ax, float ay, float etaA, float etaB, bool sepG) {
    mi_bxdf b = MakeBxDF(MI_BXDF_MICROFACET_TRANSMISSION, MI_BSDF_TRANSMISSION | MI_BSDF_GLOSSY, T);
    b.p[0] = ax; b.p[1] = ay; b.p[2] = etaA; b.p[3] = etaB; b.p[5] = sepG ? 1.f : 0.f;
    return b;
}

// A roughness parameter: a constant, or a float image texture evaluated at the hit (mi_material.rough_tex).
struct RoughSrc {
    float value = 0.f;
    int tex = -1;
    bool given = false;
};
RoughSrc Rough(const TextureParams &mp, const std::string &name, float def) {
    RoughSrc r;
    r.tex = mp.GetFloatImageTexture(name);
    if (r.tex >= 0) { r.given = true; return r; }
    float f;
    r.given = mp.GetFloatOrNull(name, &f);
    r.value = r.given ? f : def;
    return r;
}
// the alpha pair of a microfacet lobe from its two roughness sources: constants go into the lobe (through
// RoughnessToAlpha when the material remaps), textures into the material's rough_tex
void SetRough(mi_material *m, mi_bxdf *b, const RoughSrc &u, const RoughSrc &v, bool remap) {
    b->p[0] = u.tex >= 0 ? 0.f : (remap ? RoughnessToAlpha(u.value) : u.value);
    b->p[1] = v.tex >= 0 ? 0.f : (remap ? RoughnessToAlpha(v.value) : v.value);
    if (u.tex >= 0) m->rough_tex[0] = u.tex;
    if (v.tex >= 0) m->rough_tex[1] = v.tex;
    if ((u.tex >= 0 || v.tex >= 0) && remap) m->rough_flags |= MI_ROUGH_REMAP;
}

}  // namespace

bool CompileMaterial(const std::string &type, const TextureParams &mp, mi_material *m,
                     std::vector<std::string> *warnings, std::vector<std::string> *errs) {
    *m = mi_material{};
    m->eta = 1.f;
    for (int i = 0; i < MI_MAX_BXDFS; ++i) m->tex[i] = mi_lobe_tex{-1, -1, 0u, MI_LOBE_IF_R};
    m->rough_tex[0] = m->rough_tex[1] = -1;
    m->sigma_tex = -1;
    m->bump_tex = mp.GetFloatImageTexture("bumpmap");   // `if (bumpMap) Bump(bumpMap, si)`, first line of every ComputeScatteringFunctions
    if (m->bump_tex >= 0) {
        if (type == "mix" || type == "disney" || type == "metal") { errs->push_back("\"bumpmap\" on a \"" + type + "\" material is outside the hot-path scope"); return false; }
    } else {
        float bump;
        if (mp.GetFloatOrNull("bumpmap", &bump))
            warnings->push_back("constant \"bumpmap\" ignored (it displaces along the normal by a constant; only image bump maps are evaluated on this path)");
    }

    if (type == "matte") {  // src/materials/matte.cpp:45-62,64-71
        m->kind = 0;
        const SpectrumParam Kd = mp.GetSpectrumParam("Kd", Spectrum(0.5f));
        Spectrum r = Kd.s.Clamp();
        const int sigTex = mp.GetFloatImageTexture("sigma");   // (a float map: Lambertian or Oren-Nayar is decided at the hit, mi_material.sigma_tex)
        float sig = sigTex >= 0 ? 0.f : Clamp(mp.GetFloat("sigma", 0.f), 0, 90);
        if (MayBeNonBlack(Kd, r)) {
            if (sigTex >= 0) { Add(m, OrenNayar(r, 0.f), errs); m->sigma_tex = sigTex; m->textured = 1; }
            else if (sig == 0) Add(m, Lambertian(r), errs);
            else Add(m, OrenNayar(r, sig), errs);
            Bind(m, Kd.tex, Kd.scaled);
        }
        return true;
    }
    if (type == "plastic") {  // src/materials/plastic.cpp:45-70,72-84
        m->kind = 1;
        const SpectrumParam Kd = mp.GetSpectrumParam("Kd", Spectrum(0.25f)), Ks = mp.GetSpectrumParam("Ks", Spectrum(0.25f));
        Spectrum kd = Kd.s.Clamp();
        Spectrum ks = Ks.s.Clamp();
        const RoughSrc rough = Rough(mp, "roughness", .1f);
        bool remap = mp.FindBool("remaproughness", true);
        if (MayBeNonBlack(Kd, kd)) { Add(m, Lambertian(kd), errs); Bind(m, Kd.tex, Kd.scaled); }
        if (MayBeNonBlack(Ks, ks)) {
            // FresnelDielectric(1.5f, 1.f): plastic.cpp:59
            mi_bxdf b = MicrofacetReflectionDielectric(ks, 0.f, 0.f, 1.5f, 1.f);
            SetRough(m, &b, rough, rough, remap);
            Add(m, b, errs);
            Bind(m, Ks.tex, Ks.scaled);
        }
        return true;
    }
    if (type == "mirror") {  // src/materials/mirror.cpp: SpecularReflection(R, FresnelNoOp)
        m->kind = 5;
        const SpectrumParam Kr = mp.GetSpectrumParam("Kr", Spectrum(0.9f));
        Spectrum R = Kr.s.Clamp();
        if (MayBeNonBlack(Kr, R)) {
            mi_bxdf b = MakeBxDF(MI_BXDF_SPECULAR_REFLECTION, MI_BSDF_REFLECTION | MI_BSDF_SPECULAR, R);
            b.fresnel = MI_FRESNEL_NOOP;
            Add(m, b, errs);
            Bind(m, Kr.tex, Kr.scaled);
        }
        return true;
    }
    if (type == "metal") {  // src/materials/metal.cpp:58-81,128-147: MicrofacetReflection(1, TR, FresnelConductor(1, eta, k))
        m->kind = 6;
        Spectrum copperN, copperK;
        for (int i = 0; i < MI_NSPEC; ++i) { copperN.c[i] = kCopperN[i]; copperK.c[i] = kCopperK[i]; }
        const SpectrumParam pEta = mp.GetSpectrumParam("eta", copperN), pK = mp.GetSpectrumParam("k", copperK);
        if ((pEta.tex >= 0 && pEta.scaled) || (pK.tex >= 0 && pK.scaled)) { errs->push_back("metal \"eta\" / \"k\" bound to a `scale` of an image texture is outside the hot-path scope"); return false; }
        Spectrum eta = pEta.s;
        Spectrum k = pK.s;
        const RoughSrc rough = Rough(mp, "roughness", .01f), ru = Rough(mp, "uroughness", 0.f), rv = Rough(mp, "vroughness", 0.f);
        mi_bxdf b = MakeBxDF(MI_BXDF_MICROFACET_REFLECTION, MI_BSDF_REFLECTION | MI_BSDF_GLOSSY, Spectrum(1.f));
        b.fresnel = MI_FRESNEL_CONDUCTOR;
        SetRough(m, &b, ru.given ? ru : rough, rv.given ? rv : rough, mp.FindBool("remaproughness", true));   // metal.cpp:66-73
        SetS(b, eta);
        for (int i = 0; i < MI_NSPEC; ++i) b.K[i] = k.c[i];
        Add(m, b, errs);
        Bind(m, pK.tex, false, pEta.tex, false, MI_LOBE_METAL);   // (tex_R carries k's texture: mi_lobe_rule)
        return true;
    }
    if (type == "substrate") {  // src/materials/substrate.cpp:45-64,66-81: FresnelBlend(Kd, Ks, TR)
        m->kind = 7;
        const SpectrumParam Kd = mp.GetSpectrumParam("Kd", Spectrum(.5f)), Ks = mp.GetSpectrumParam("Ks", Spectrum(.5f));
        Spectrum d = Kd.s.Clamp();
        Spectrum s = Ks.s.Clamp();
        const RoughSrc roughu = Rough(mp, "uroughness", .1f), roughv = Rough(mp, "vroughness", .1f);
        if (MayBeNonBlack(Kd, d) || MayBeNonBlack(Ks, s)) {
            mi_bxdf b = MakeBxDF(MI_BXDF_FRESNEL_BLEND, MI_BSDF_REFLECTION | MI_BSDF_GLOSSY, d);
            SetS(b, s);
            SetRough(m, &b, roughu, roughv, mp.FindBool("remaproughness", true));
            Add(m, b, errs);
            Bind(m, Kd.tex, Kd.scaled, Ks.tex, Ks.scaled, MI_LOBE_IF_R_OR_S);
        }
        return true;
    }
    if (type == "translucent") {  // src/materials/translucent.cpp:45-83,85-102
        m->kind = 8;
        const float eta = 1.5f;
        m->eta = eta;
        Spectrum r = mp.GetSpectrum("reflect", Spectrum(0.5f)).Clamp();
        Spectrum t = mp.GetSpectrum("transmit", Spectrum(0.5f)).Clamp();
        if (r.IsBlack() && t.IsBlack()) return true;
        // (a textured Kd / Ks: the lobe keeps r or t as its constant and the texture value multiplies it at the hit)
        const SpectrumParam Kd = mp.GetSpectrumParam("Kd", Spectrum(0.25f));
        Spectrum kd = Kd.s.Clamp();
        if (MayBeNonBlack(Kd, kd)) {
            if (!r.IsBlack()) { Add(m, Lambertian(r * kd), errs); Bind(m, Kd.tex, true, -1, false, MI_LOBE_IF_TEX); }
            if (!t.IsBlack()) {
                Add(m, MakeBxDF(MI_BXDF_LAMBERTIAN_TRANSMISSION, MI_BSDF_TRANSMISSION | MI_BSDF_DIFFUSE, t * kd), errs);
                Bind(m, Kd.tex, true, -1, false, MI_LOBE_IF_TEX);
            }
        }
        const SpectrumParam Ks = mp.GetSpectrumParam("Ks", Spectrum(0.25f));
        Spectrum ks = Ks.s.Clamp();
        if (MayBeNonBlack(Ks, ks) && (!r.IsBlack() || !t.IsBlack())) {
            const RoughSrc rough = Rough(mp, "roughness", .1f);
            const bool remap = mp.FindBool("remaproughness", true);
            if (!r.IsBlack()) {
                mi_bxdf b = MicrofacetReflectionDielectric(r * ks, 0.f, 0.f, 1.f, eta);
                SetRough(m, &b, rough, rough, remap);
                Add(m, b, errs); Bind(m, Ks.tex, true, -1, false, MI_LOBE_IF_TEX);
            }
            if (!t.IsBlack()) {
                mi_bxdf b = MicrofacetTransmission(t * ks, 0.f, 0.f, 1.f, eta, false);
                SetRough(m, &b, rough, rough, remap);
                Add(m, b, errs); Bind(m, Ks.tex, true, -1, false, MI_LOBE_IF_TEX);
            }
        }
        return true;
    }
    if (type == "glass") {  // src/materials/glass.cpp:45-92,94-111 (allowMultipleLobes = true)
        m->kind = 2;
        const SpectrumParam Kr = mp.GetSpectrumParam("Kr", Spectrum(1.f)), Kt = mp.GetSpectrumParam("Kt", Spectrum(1.f));
        Spectrum R = Kr.s.Clamp();
        Spectrum T = Kt.s.Clamp();
        float eta;
        if (!mp.GetFloatOrNull("eta", &eta)) eta = mp.GetFloat("index", 1.5f);
        const RoughSrc ruSrc = Rough(mp, "uroughness", 0.f), rvSrc = Rough(mp, "vroughness", 0.f);
        float urough = ruSrc.value;
        float vrough = rvSrc.value;
        bool remap = mp.FindBool("remaproughness", true);
        m->eta = eta;
        if (!MayBeNonBlack(Kr, R) && !MayBeNonBlack(Kt, T)) return true;
        if (ruSrc.tex >= 0 || rvSrc.tex >= 0) {
            // a roughness map: whether the surface is the specular or the rough glass is decided at the hit (MI_ROUGH_GLASS)
            if (Kr.tex >= 0 || Kt.tex >= 0) { errs->push_back("\"glass\" with a roughness map and image-textured Kr / Kt is outside the hot-path scope"); return false; }
            mi_bxdf sp = MakeBxDF(MI_BXDF_FRESNEL_SPECULAR, MI_BSDF_REFLECTION | MI_BSDF_TRANSMISSION | MI_BSDF_SPECULAR, R);
            SetS(sp, T);
            sp.p[0] = 1.f; sp.p[1] = eta;
            sp.p[6] = ruSrc.value; sp.p[7] = rvSrc.value;
            Add(m, sp, errs);
            if (!R.IsBlack()) { mi_bxdf b = MicrofacetReflectionDielectric(R, 0.f, 0.f, 1.f, eta); SetRough(m, &b, ruSrc, rvSrc, remap); Add(m, b, errs); }
            if (!T.IsBlack()) { mi_bxdf b = MicrofacetTransmission(T, 0.f, 0.f, 1.f, eta, false); SetRough(m, &b, ruSrc, rvSrc, remap); Add(m, b, errs); }
            m->rough_flags |= MI_ROUGH_GLASS;
            m->textured = 1;
            return true;
        }
        bool isSpecular = urough == 0 && vrough == 0;
        if (isSpecular) {
            mi_bxdf b = MakeBxDF(MI_BXDF_FRESNEL_SPECULAR,
                                 MI_BSDF_REFLECTION | MI_BSDF_TRANSMISSION | MI_BSDF_SPECULAR, R);
            SetS(b, T);
            b.p[0] = 1.f; b.p[1] = eta;
            Add(m, b, errs);
            Bind(m, Kr.tex, Kr.scaled, Kt.tex, Kt.scaled, MI_LOBE_IF_R_OR_S);
        } else {
            if ((Kr.tex >= 0) != (Kt.tex >= 0) || (Kr.tex >= 0 && Kt.tex >= 0)) {
                // rough glass adds its lobes only `if (R.IsBlack() && T.IsBlack()) return` has not fired (glass.cpp:70-72):
                // with one side textured that early-out couples the two lobes, which mi_lobe_tex does not express
                errs->push_back("rough \"glass\" with image-textured Kr / Kt is outside the hot-path scope");
                return false;
            }
            if (remap) { urough = RoughnessToAlpha(urough); vrough = RoughnessToAlpha(vrough); }
            if (!R.IsBlack()) Add(m, MicrofacetReflectionDielectric(R, urough, vrough, 1.f, eta), errs);
            if (!T.IsBlack()) Add(m, MicrofacetTransmission(T, urough, vrough, 1.f, eta, false), errs);
        }
        return true;
    }
    if (type == "uber") {  // src/materials/uber.cpp:45-101,103-128
        m->kind = 3;
        const SpectrumParam pKd = mp.GetSpectrumParam("Kd", Spectrum(0.25f)), pKs = mp.GetSpectrumParam("Ks", Spectrum(0.25f));
        const SpectrumParam pKr = mp.GetSpectrumParam("Kr", Spectrum(0.f)), pKt = mp.GetSpectrumParam("Kt", Spectrum(0.f));
        const Spectrum Kd = pKd.s, Ks = pKs.s, Kr = pKr.s, Kt = pKt.s;   // (textured: 1, so that op * K below leaves op as the lobe's constant)
        const RoughSrc roughness = Rough(mp, "roughness", .1f), ru = Rough(mp, "uroughness", 0.f), rv = Rough(mp, "vroughness", 0.f);
        float e;
        if (!mp.GetFloatOrNull("eta", &e)) e = mp.GetFloat("index", 1.5f);
        Spectrum opacity = mp.GetSpectrum("opacity", Spectrum(1.f));
        bool remap = mp.FindBool("remaproughness", true);

        Spectrum op = opacity.Clamp();
        Spectrum t = (-op + Spectrum(1.f)).Clamp();
        if (!t.IsBlack()) {
            m->eta = 1.f;
            Add(m, SpecularTransmission(t, 1.f, 1.f), errs);
        } else
            m->eta = e;
        Spectrum kd = op * Kd.Clamp();
        if (!kd.IsBlack()) { Add(m, Lambertian(kd), errs); Bind(m, pKd.tex, true); }
        Spectrum ks = op * Ks.Clamp();
        if (!ks.IsBlack()) {
            const RoughSrc roughu = ru.given ? ru : roughness;   // uber.cpp:88-96
            const RoughSrc roughv = rv.given ? rv : roughu;
            mi_bxdf b = MicrofacetReflectionDielectric(ks, 0.f, 0.f, 1.f, e);
            SetRough(m, &b, roughu, roughv, remap);
            Add(m, b, errs);
            Bind(m, pKs.tex, true);
        }
        Spectrum kr = op * Kr.Clamp();
        if (!kr.IsBlack()) { Add(m, SpecularReflectionDielectric(kr, 1.f, e), errs); Bind(m, pKr.tex, true); }
        Spectrum kt = op * Kt.Clamp();
        if (!kt.IsBlack()) { Add(m, SpecularTransmission(kt, 1.f, e), errs); Bind(m, pKt.tex, true); }
        return true;
    }
    if (type == "disney") {  // src/materials/disney.cpp:474-587,589-624
        m->kind = 4;
        // (an image-textured "color": the lobes keep their weights as constants -- c = 1 below -- and take the colour at
        // the hit, each by its rule: mi_lobe_rule, MI_LOBE_ALWAYS ... MI_LOBE_DISNEY_STRANS)
        const SpectrumParam pc = mp.GetSpectrumParam("color", Spectrum(0.5f));
        const int ctex = pc.tex;
        if (ctex >= 0 && pc.scaled) { errs->push_back("disney \"color\" bound to a `scale` of an image texture is outside the hot-path scope"); return false; }
        Spectrum c = ctex >= 0 ? Spectrum(1.f) : pc.s.Clamp();
        float metallicWeight = mp.GetFloat("metallic", 0.f);
        float e = mp.GetFloat("eta", 1.5f);
        const RoughSrc roughSrc = Rough(mp, "roughness", .5f);   // (a float map: MI_ROUGH_DISNEY)
        float rough = roughSrc.value;
        float specTint = mp.GetFloat("speculartint", 0.f);
        float anisotropic = mp.GetFloat("anisotropic", 0.f);
        float sheenWeight = mp.GetFloat("sheen", 0.f);
        float stint = mp.GetFloat("sheentint", .5f);
        float cc = mp.GetFloat("clearcoat", 0.f);
        float ccGloss = mp.GetFloat("clearcoatgloss", 1.f);
        float strans = mp.GetFloat("spectrans", 0.f);
        Spectrum sd = mp.GetSpectrum("scatterdistance", Spectrum(0.));
        bool thin = mp.FindBool("thin", false);
        float flat = mp.GetFloat("flatness", 0.f);
        float dtIn = mp.GetFloat("difftrans", 1.f);

        float diffuseWeight = (1 - metallicWeight) * (1 - strans);
        float dt = dtIn / 2;
        float lum = c.y();
        Spectrum Ctint = lum > 0 ? (c / lum) : Spectrum(1.);
        Spectrum Csheen;
        if (sheenWeight > 0) Csheen = Lerp(stint, Spectrum(1.), Ctint);
        const int diffFlags = MI_BSDF_REFLECTION | MI_BSDF_DIFFUSE;
        if (diffuseWeight > 0) {
            if (thin) {
                Add(m, MakeBxDF(MI_BXDF_DISNEY_DIFFUSE, diffFlags, diffuseWeight * (1 - flat) * (1 - dt) * c), errs);
                Bind(m, ctex, true, -1, false, MI_LOBE_ALWAYS);
                mi_bxdf ss = MakeBxDF(MI_BXDF_DISNEY_FAKE_SS, diffFlags, diffuseWeight * flat * (1 - dt) * c);
                ss.p[0] = rough;
                Add(m, ss, errs);
                Bind(m, ctex, true, -1, false, MI_LOBE_ALWAYS);
            } else {
                if (sd.IsBlack()) {
                    Add(m, MakeBxDF(MI_BXDF_DISNEY_DIFFUSE, diffFlags, diffuseWeight * c), errs);
                    Bind(m, ctex, true, -1, false, MI_LOBE_ALWAYS);
                } else {
                    errs->push_back("disney \"scatterdistance\" (BSSRDF) is outside the hot-path scope (SURVEY 2 row 6)");
                    return false;
                }
            }
            mi_bxdf retro = MakeBxDF(MI_BXDF_DISNEY_RETRO, diffFlags, diffuseWeight * c);
            retro.p[0] = rough;
            Add(m, retro, errs);
            Bind(m, ctex, true, -1, false, MI_LOBE_ALWAYS);
            if (sheenWeight > 0) {
                mi_bxdf sh = MakeBxDF(MI_BXDF_DISNEY_SHEEN, diffFlags, diffuseWeight * sheenWeight * Csheen);
                sh.p[6] = diffuseWeight * sheenWeight; sh.p[7] = stint;
                Add(m, sh, errs);
                Bind(m, ctex, false, -1, false, MI_LOBE_DISNEY_SHEEN);
            }
        }
        // "1 - anisotropic * .9" is evaluated in double, then sqrt(double) -> Float
        float aspect = (float)std::sqrt(1 - (double)anisotropic * .9);
        float ax = std::max(float(.001), sqr(rough) / aspect);
        float ay = std::max(float(.001), sqr(rough) * aspect);
        float r0 = sqr(e - 1) / sqr(e + 1);  // SchlickR0FromEta
        Spectrum Cspec0 = Lerp(metallicWeight, r0 * Lerp(specTint, Spectrum(1.), Ctint), c);
        {
            mi_bxdf b = MakeBxDF(MI_BXDF_MICROFACET_REFLECTION, MI_BSDF_REFLECTION | MI_BSDF_GLOSSY, c);
            b.fresnel = MI_FRESNEL_DISNEY;
            SetS(b, Cspec0);
            b.p[0] = ax; b.p[1] = ay; b.p[2] = metallicWeight; b.p[3] = e; b.p[5] = 1.f;
            b.p[4] = aspect; b.p[6] = specTint; b.p[7] = r0;
            Add(m, b, errs);
            Bind(m, ctex, false, ctex, false, MI_LOBE_DISNEY_SPEC);
        }
        if (cc > 0) {
            mi_bxdf b = MakeBxDF(MI_BXDF_DISNEY_CLEARCOAT, MI_BSDF_REFLECTION | MI_BSDF_GLOSSY, Spectrum(0.f));
            b.p[0] = cc;
            // Lerp(gloss, .1, .001): double literals converted to Float arguments
            b.p[1] = Lerp(ccGloss, (float).1, (float).001);
            Add(m, b, errs);
        }
        if (strans > 0) {
            Spectrum T = strans * Sqrt(c);
            mi_bxdf tb;
            if (thin) {
                float rscaled = (0.65f * e - 0.35f) * rough;
                float ax2 = std::max(float(.001), sqr(rscaled) / aspect);
                float ay2 = std::max(float(.001), sqr(rscaled) * aspect);
                tb = MicrofacetTransmission(T, ax2, ay2, 1.f, e, false);
                tb.p[7] = 0.65f * e - 0.35f;
            } else
                tb = MicrofacetTransmission(T, ax, ay, 1.f, e, true);
            tb.p[4] = aspect; tb.p[6] = strans;
            Add(m, tb, errs);
            Bind(m, ctex, false, -1, false, MI_LOBE_DISNEY_STRANS);
        }
        if (thin) {
            Add(m, MakeBxDF(MI_BXDF_LAMBERTIAN_TRANSMISSION, MI_BSDF_TRANSMISSION | MI_BSDF_DIFFUSE, dt * c), errs);
            Bind(m, ctex, true, -1, false, MI_LOBE_ALWAYS);
        }
        if (roughSrc.tex >= 0) { m->rough_tex[0] = roughSrc.tex; m->rough_flags |= MI_ROUGH_DISNEY; m->textured = 1; }
        return errs->empty() || m->n_bxdfs <= MI_MAX_BXDFS;
    }
    errs->push_back("material \"" + type + "\" is outside the PathIntegrator hot-path scope (SURVEY 2 row 18)");
    return false;
}

// MixMaterial::ComputeScatteringFunctions (src/materials/mixmat.cpp:46-64): the lobes of m1 wrapped in
// ScaledBxDF(s1 = amount.Clamp()), then those of m2 in ScaledBxDF(s2 = (1 - s1).Clamp()); the BSDF (and
// its eta) is m1's.
bool CompileMixMaterial(const mi_material &m1, const mi_material &m2, const Spectrum &amount, mi_material *out,
                        std::vector<std::string> *errs) {
    *out = mi_material{};
    for (int i = 0; i < MI_MAX_BXDFS; ++i) out->tex[i] = mi_lobe_tex{-1, -1, 0u, MI_LOBE_IF_R};
    // Bump maps (mixmat.cpp:52-56): m1 bumps the interaction in place and builds the BSDF -- whose frame the mixed lobes share --, m2 gets
    // a copy of that interaction and its BSDF's lobes only: m1's bump map shapes the shading frame, m2's has no effect on anything.
    out->bump_tex = m1.bump_tex;
    out->rough_tex[0] = out->rough_tex[1] = -1;
    out->sigma_tex = -1;
    out->kind = 9;
    out->eta = m1.eta;
    const Spectrum s1 = amount.Clamp();
    const Spectrum s2 = (Spectrum(1.f) - s1).Clamp();
    const mi_material *src[2] = {&m1, &m2};
    const Spectrum *sc[2] = {&s1, &s2};
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < src[k]->n_bxdfs; ++i) {
            mi_bxdf b = src[k]->bxdf[i];
            // (each sub-material runs its own ComputeScatteringFunctions, mixmat.cpp:52-56: its textured lobes keep their
            // bindings and presence rules; a bump or roughness map would act on one sub-material's copy of the interaction)
            // a roughness or sigma map acts on one sub-material's lobes only, which the per-material overrides do not express
            if (src[k]->rough_tex[0] >= 0 || src[k]->rough_tex[1] >= 0 || src[k]->sigma_tex >= 0) {
                errs->push_back("a \"mix\" of roughness-mapped or sigma-mapped materials is outside the hot-path scope");
                return false;
            }
            if (b.scaled >= 2) { errs->push_back("a \"mix\" of a \"mix\" of \"mix\" materials (three nested ScaledBxDFs) is not built on this path"); return false; }
            if (b.scaled == 1) { b.scaled = 2; for (int j = 0; j < MI_NSPEC; ++j) b.scale2[j] = sc[k]->c[j]; }   // ScaledBxDF(ScaledBxDF(lobe, inner), outer)
            else { b.scaled = 1; for (int j = 0; j < MI_NSPEC; ++j) b.scale[j] = sc[k]->c[j]; }
            if (!Add(out, b, errs)) return false;
            out->tex[out->n_bxdfs - 1] = src[k]->tex[i];
            if (src[k]->tex[i].tex_R >= 0 || src[k]->tex[i].tex_S >= 0) out->textured = 1;
        }
    return true;
}

}  // namespace mipt
