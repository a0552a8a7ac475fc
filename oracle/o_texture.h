// oracle/o_texture.h -- TEST INFRASTRUCTURE (CPU oracle).
// ImageTexture<RGBSpectrum, Spectrum>::Evaluate over the host-built pyramid of include/mi_pt.h (mi_mipmap /
// mi_texture), restating
//   SurfaceInteraction::ComputeDifferentials   src/core/interaction.cpp:99-143
//   UVMapping2D::Map                           src/core/texture.cpp:91-99
//   MIPMap::Lookup (trilinear / EWA), triangle, EWA, Texel   src/core/mipmap.h:213-385
//   convertOut + SampledSpectrum::FromRGB (Illuminant, its default)   src/textures/imagemap.h:113-117, src/core/spectrum.cpp:98-180
#pragma once
#include "../include/mi_pt.h"
#include "o_math.h"
#include "o_shapes.h"

namespace orc {

struct RayDifferential {  // the offset rays of geometry.h:897-931 (the main ray travels separately)
    bool hasDifferentials = false;
    V3 rxOrigin, ryOrigin, rxDirection, ryDirection;
    void ScaleDifferentials(const V3 &o, const V3 &d, Float s) {  // geometry.h:917-922
        rxOrigin = o + (rxOrigin - o) * s;
        ryOrigin = o + (ryOrigin - o) * s;
        rxDirection = d + (rxDirection - d) * s;
        ryDirection = d + (ryDirection - d) * s;
    }
};

struct TexDifferentials { Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0; };

inline bool SolveLinearSystem2x2(const Float A[2][2], const Float B[2], Float *x0, Float *x1) {  // transform.cpp:41-49
    Float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    if (std::abs(det) < 1e-10f) return false;
    *x0 = (A[1][1] * B[0] - A[0][1] * B[1]) / det;
    *x1 = (A[0][0] * B[1] - A[1][0] * B[0]) / det;
    if (std::isnan(*x0) || std::isnan(*x1)) return false;
    return true;
}

inline TexDifferentials ComputeDifferentials(const SurfaceInteraction &si, const RayDifferential &ray) {
    TexDifferentials t;
    if (!ray.hasDifferentials) return t;
    const V3 &n = si.n, &p = si.p;
    Float d = Dot(n, V3(p.x, p.y, p.z));
    Float tx = -(Dot(n, V3(ray.rxOrigin)) - d) / Dot(n, ray.rxDirection);
    if (std::isinf(tx) || std::isnan(tx)) return t;
    V3 px = ray.rxOrigin + tx * ray.rxDirection;
    Float ty = -(Dot(n, V3(ray.ryOrigin)) - d) / Dot(n, ray.ryDirection);
    if (std::isinf(ty) || std::isnan(ty)) return t;
    V3 py = ray.ryOrigin + ty * ray.ryDirection;
    int dim[2];
    if (std::abs(n.x) > std::abs(n.y) && std::abs(n.x) > std::abs(n.z)) { dim[0] = 1; dim[1] = 2; }
    else if (std::abs(n.y) > std::abs(n.z)) { dim[0] = 0; dim[1] = 2; }
    else { dim[0] = 0; dim[1] = 1; }
    Float A[2][2] = {{si.dpdu[dim[0]], si.dpdv[dim[0]]}, {si.dpdu[dim[1]], si.dpdv[dim[1]]}};
    Float Bx[2] = {px[dim[0]] - p[dim[0]], px[dim[1]] - p[dim[1]]};
    Float By[2] = {py[dim[0]] - p[dim[0]], py[dim[1]] - p[dim[1]]};
    if (!SolveLinearSystem2x2(A, Bx, &t.dudx, &t.dvdx)) t.dudx = t.dvdx = 0;
    if (!SolveLinearSystem2x2(A, By, &t.dudy, &t.dvdy)) t.dudy = t.dvdy = 0;
    return t;
}

struct RGB3 {
    Float c[3];
    RGB3(Float v = 0.f) { c[0] = c[1] = c[2] = v; }
    RGB3 operator+(const RGB3 &o) const { RGB3 r; for (int i = 0; i < 3; ++i) r.c[i] = c[i] + o.c[i]; return r; }
    RGB3 &operator+=(const RGB3 &o) { for (int i = 0; i < 3; ++i) c[i] += o.c[i]; return *this; }
    RGB3 operator*(Float a) const { RGB3 r; for (int i = 0; i < 3; ++i) r.c[i] = c[i] * a; return r; }
    RGB3 operator/(Float a) const { RGB3 r; for (int i = 0; i < 3; ++i) r.c[i] = c[i] / a; return r; }  // spectrum.h:187-194
};
inline RGB3 operator*(Float a, const RGB3 &s) { return s * a; }

struct MipView {
    const mi_mipmap &m;
    int Levels() const { return m.n_levels; }
    int uSize(int level) const { return std::max(1, m.width >> level); }
    int vSize(int level) const { return std::max(1, m.height >> level); }
    static int Mod(int a, int b) { int r = a - (a / b) * b; return (r < 0) ? r + b : r; }
    RGB3 Texel(int level, int s, int t) const {  // mipmap.h:213-235
        const int w = uSize(level), h = vSize(level);
        switch (m.wrap) {
        case 0: s = Mod(s, w); t = Mod(t, h); break;
        case 2: s = Clamp(s, 0, w - 1); t = Clamp(t, 0, h - 1); break;
        default: if (s < 0 || s >= w || t < 0 || t >= h) return RGB3(0.f); break;
        }
        const float *px = m.texels + 3 * ((size_t)m.level_offset[level] + (size_t)t * w + s);
        RGB3 r;
        r.c[0] = px[0]; r.c[1] = px[1]; r.c[2] = px[2];
        return r;
    }
    RGB3 triangle(int level, const Float st[2]) const {  // mipmap.h:268-279
        level = Clamp(level, 0, Levels() - 1);
        Float s = st[0] * uSize(level) - 0.5f;
        Float t = st[1] * vSize(level) - 0.5f;
        int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
        Float ds = s - s0, dt = t - t0;
        return (1 - ds) * (1 - dt) * Texel(level, s0, t0) + (1 - ds) * dt * Texel(level, s0, t0 + 1) +
               ds * (1 - dt) * Texel(level, s0 + 1, t0) + ds * dt * Texel(level, s0 + 1, t0 + 1);
    }
    static Float Log2(Float x) { const Float invLog2 = 1.442695040888963387004650940071; return LogF(x) * invLog2; }
    RGB3 LookupWidth(const Float st[2], Float width, bool noFiltering) const {  // mipmap.h:238-266
        if (noFiltering) {
            Float s = st[0] * uSize(0) - 0.5f;
            Float t = st[1] * vSize(0) - 0.5f;
            int s0 = (int)std::round(s), t0 = (int)std::round(t);
            return Texel(0, s0, t0);
        }
        Float level = Levels() - 1 + Log2(std::max(width, (Float)1e-8));
        if (level < 0) return triangle(0, st);
        else if (level >= Levels() - 1) return Texel(Levels() - 1, 0, 0);
        int iLevel = (int)std::floor(level);
        Float delta = level - iLevel;
        return (1 - delta) * triangle(iLevel, st) + delta * triangle(iLevel + 1, st);
    }
    static const Float *WeightLut() {  // mipmap.h:199-206
        static Float lut[128];
        static bool init = false;
        if (!init) {
            for (int i = 0; i < 128; ++i) {
                Float alpha = 2;
                Float r2 = Float(i) / Float(128 - 1);
                lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
            }
            init = true;
        }
        return lut;
    }
    RGB3 EWA(int level, const Float stIn[2], const Float d0[2], const Float d1[2]) const {  // mipmap.h:321-380
        if (level >= Levels()) return Texel(Levels() - 1, 0, 0);
        Float st[2] = {stIn[0] * uSize(level) - 0.5f, stIn[1] * vSize(level) - 0.5f};
        Float dst0[2] = {d0[0] * uSize(level), d0[1] * vSize(level)};
        Float dst1[2] = {d1[0] * uSize(level), d1[1] * vSize(level)};
        Float A = dst0[1] * dst0[1] + dst1[1] * dst1[1] + 1;
        Float B = -2 * (dst0[0] * dst0[1] + dst1[0] * dst1[1]);
        Float C = dst0[0] * dst0[0] + dst1[0] * dst1[0] + 1;
        Float invF = 1 / (A * C - B * B * 0.25f);
        A *= invF; B *= invF; C *= invF;
        Float det = -B * B + 4 * A * C;
        Float invDet = 1 / det;
        Float uSqrt = std::sqrt(det * C), vSqrt = std::sqrt(A * det);
        int s0 = (int)std::ceil(st[0] - 2 * invDet * uSqrt);
        int s1 = (int)std::floor(st[0] + 2 * invDet * uSqrt);
        int t0 = (int)std::ceil(st[1] - 2 * invDet * vSqrt);
        int t1 = (int)std::floor(st[1] + 2 * invDet * vSqrt);
        RGB3 sum(0.f);
        Float sumWts = 0;
        const Float *weightLut = WeightLut();
        for (int it = t0; it <= t1; ++it) {
            Float tt = it - st[1];
            for (int is = s0; is <= s1; ++is) {
                Float ss = is - st[0];
                Float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
                if (r2 < 1) {
                    int index = std::min((int)(r2 * 128), 128 - 1);
                    Float weight = weightLut[index];
                    sum += Texel(level, is, it) * weight;
                    sumWts += weight;
                }
            }
        }
        return sum / sumWts;
    }
    RGB3 Lookup(const Float st[2], const Float dstdx[2], const Float dstdy[2], int filter, Float maxAnisotropy) const {  // mipmap.h:281-319
        if (filter != MI_TEX_EWA) {
            Float width = std::max(std::max(std::abs(dstdx[0]), std::abs(dstdx[1])), std::max(std::abs(dstdy[0]), std::abs(dstdy[1])));
            return LookupWidth(st, 2 * width, filter == MI_TEX_NONE);
        }
        Float dst0[2] = {dstdx[0], dstdx[1]}, dst1[2] = {dstdy[0], dstdy[1]};
        if (dst0[0] * dst0[0] + dst0[1] * dst0[1] < dst1[0] * dst1[0] + dst1[1] * dst1[1]) { std::swap(dst0[0], dst1[0]); std::swap(dst0[1], dst1[1]); }
        Float majorLength = std::sqrt(dst0[0] * dst0[0] + dst0[1] * dst0[1]);
        Float minorLength = std::sqrt(dst1[0] * dst1[0] + dst1[1] * dst1[1]);
        if (minorLength * maxAnisotropy < majorLength && minorLength > 0) {
            Float scale = majorLength / (minorLength * maxAnisotropy);
            dst1[0] *= scale; dst1[1] *= scale;
            minorLength *= scale;
        }
        if (minorLength == 0) return triangle(0, st);
        Float lod = std::max((Float)0, Levels() - (Float)1 + Log2(minorLength));
        int ilod = (int)std::floor(lod);
        Float t = lod - ilod;
        return (1 - t) * EWA(ilod, st, dst0, dst1) + t * EWA(ilod + 1, st, dst0, dst1);
    }
};

// SampledSpectrum::FromRGB(rgb) -- the default SpectrumType is Illuminant (spectrum.h:428-429), which is what
// ImageTexture::convertOut (imagemap.h:113-117) and InfiniteAreaLight use; spectrum.cpp:98-180
inline Spec SpecFromRGBIllum(const mi_scene_desc &d, const Float rgb[3]) {
    const Spec white = Spec::From(d.rgb_illum[0]), cyan = Spec::From(d.rgb_illum[1]), magenta = Spec::From(d.rgb_illum[2]),
               yellow = Spec::From(d.rgb_illum[3]), red = Spec::From(d.rgb_illum[4]), green = Spec::From(d.rgb_illum[5]),
               blue = Spec::From(d.rgb_illum[6]);
    Spec r;
    if (rgb[0] <= rgb[1] && rgb[0] <= rgb[2]) {
        r += rgb[0] * white;
        if (rgb[1] <= rgb[2]) { r += (rgb[1] - rgb[0]) * cyan; r += (rgb[2] - rgb[1]) * blue; }
        else { r += (rgb[2] - rgb[0]) * cyan; r += (rgb[1] - rgb[2]) * green; }
    } else if (rgb[1] <= rgb[0] && rgb[1] <= rgb[2]) {
        r += rgb[1] * white;
        if (rgb[0] <= rgb[2]) { r += (rgb[0] - rgb[1]) * magenta; r += (rgb[2] - rgb[0]) * blue; }
        else { r += (rgb[2] - rgb[1]) * magenta; r += (rgb[0] - rgb[2]) * red; }
    } else {
        r += rgb[2] * white;
        if (rgb[0] <= rgb[1]) { r += (rgb[0] - rgb[2]) * yellow; r += (rgb[1] - rgb[0]) * green; }
        else { r += (rgb[1] - rgb[2]) * yellow; r += (rgb[0] - rgb[1]) * red; }
    }
    r *= .86445f;
    for (int i = 0; i < NS; ++i) r.c[i] = Clamp(r.c[i], 0, Infinity);
    return r;
}

inline Float AlphaTextureValue(const mi_scene_desc &d, int tex, Float u, Float v) {
    const mi_texture &t = d.textures[tex];
    const Float zero[2] = {0, 0};
    const Float st[2] = {t.su * u + t.du, t.sv * v + t.dv};
    MipView mip{d.mipmaps[t.mipmap]};
    return mip.Lookup(st, zero, zero, t.filter, t.max_aniso).c[0] * t.post_scale;
}

// Texture<Float>::Evaluate(si) of float image texture `tex` at (u, v) with the hit's differentials
inline Float EvalFloatImageTexture(const mi_scene_desc &d, int tex, Float u, Float v, const TexDifferentials &td) {
    const mi_texture &t = d.textures[tex];
    const Float dstdx[2] = {t.su * td.dudx, t.sv * td.dvdx}, dstdy[2] = {t.su * td.dudy, t.sv * td.dvdy};
    const Float st[2] = {t.su * u + t.du, t.sv * v + t.dv};
    MipView mip{d.mipmaps[t.mipmap]};
    return mip.Lookup(st, dstdx, dstdy, t.filter, t.max_aniso).c[0] * t.post_scale;
}

// Material::Bump (material.cpp:47-84) with a uv-mapped displacement texture: only (u, v) of the shifted evaluation
// points matters to the texture, so their positions and normals are not formed.
inline void Bump(const mi_scene_desc &d, int tex, SurfaceInteraction *si, const TexDifferentials &td) {
    Float du = .5f * (std::abs(td.dudx) + std::abs(td.dudy));
    if (du == 0) du = .0005f;
    Float uDisplace = EvalFloatImageTexture(d, tex, si->uv[0] + du, si->uv[1], td);
    Float dv = .5f * (std::abs(td.dvdx) + std::abs(td.dvdy));
    if (dv == 0) dv = .0005f;
    Float vDisplace = EvalFloatImageTexture(d, tex, si->uv[0], si->uv[1] + dv, td);
    Float displace = EvalFloatImageTexture(d, tex, si->uv[0], si->uv[1], td);
    V3 dpdu = si->shading.dpdu + (uDisplace - displace) / du * V3(si->shading.n) + displace * V3(si->shading.dndu);
    V3 dpdv = si->shading.dpdv + (vDisplace - displace) / dv * V3(si->shading.n) + displace * V3(si->shading.dndv);
    // SetShadingGeometry(dpdu, dpdv, shading.dndu, shading.dndv, false), interaction.cpp:76-93
    si->shading.n = Normalize(Cross(dpdu, dpdv));
    if (si->flip) si->shading.n = -si->shading.n;
    si->shading.n = Faceforward(si->shading.n, si->n);
    si->shading.dpdu = dpdu; si->shading.dpdv = dpdv;
}

// Texture<Spectrum>::Evaluate(si).Clamp() for image texture `tex`
// Checkerboard2DTexture<Spectrum>::Evaluate over constant tex1 / tex2, checkerboard.h:47-86: the weight of tex2
inline Float CheckerboardArea2(const Float st[2], const Float dstdx[2], const Float dstdy[2], bool aaNone) {
    const Float point = (((int)std::floor(st[0]) + (int)std::floor(st[1])) % 2 == 0) ? 0.f : 1.f;
    if (aaNone) return point;
    Float ds = std::max(std::abs(dstdx[0]), std::abs(dstdy[0]));
    Float dt = std::max(std::abs(dstdx[1]), std::abs(dstdy[1]));
    Float s0 = st[0] - ds, s1 = st[0] + ds;
    Float t0 = st[1] - dt, t1 = st[1] + dt;
    if (std::floor(s0) == std::floor(s1) && std::floor(t0) == std::floor(t1)) return point;
    auto bumpInt = [](Float x) { return (int)std::floor(x / 2) + 2 * std::max(x / 2 - (int)std::floor(x / 2) - (Float)0.5, (Float)0); };
    Float sint = (bumpInt(s1) - bumpInt(s0)) / (2 * ds);
    Float tint = (bumpInt(t1) - bumpInt(t0)) / (2 * dt);
    Float area2 = sint + tint - 2 * sint * tint;
    if (ds > 1 || dt > 1) area2 = .5f;
    return area2;
}

inline Spec EvalImageTexture(const mi_scene_desc &d, int tex, const SurfaceInteraction &si, const TexDifferentials &td) {
    const mi_texture &t = d.textures[tex];
    const Float dstdx[2] = {t.su * td.dudx, t.sv * td.dvdx}, dstdy[2] = {t.su * td.dudy, t.sv * td.dvdy};
    const Float st[2] = {t.su * si.uv[0] + t.du, t.sv * si.uv[1] + t.dv};
    if (t.type == MI_TEX_CHECKERBOARD) {
        const Float area2 = CheckerboardArea2(st, dstdx, dstdy, t.aa_none != 0);
        Spec s = (1 - area2) * Spec::From(t.spec1) + area2 * Spec::From(t.spec2);
        for (int i = 0; i < NS; ++i) s.c[i] = Clamp(s.c[i], 0, Infinity);
        return s;
    }
    MipView mip{d.mipmaps[t.mipmap]};
    const RGB3 mem = mip.Lookup(st, dstdx, dstdy, t.filter, t.max_aniso);
    Spec s = SpecFromRGBIllum(d, mem.c);
    for (int i = 0; i < NS; ++i) s.c[i] = Clamp(s.c[i], 0, Infinity);
    return s;
}

}  // namespace orc
