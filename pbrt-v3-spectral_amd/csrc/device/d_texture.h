// d_texture.h -- image textures at a path vertex (ABI v5: mi_texture / mi_mipmap / mi_lobe_tex).
//   SurfaceInteraction::ComputeDifferentials          src/core/interaction.cpp:99-143
//   UVMapping2D::Map                                  src/core/texture.cpp:91-99
//   MIPMap::Lookup (trilinear, EWA), triangle, Texel  src/core/mipmap.h:213-385
//   convertOut + SampledSpectrum::FromRGB (Illuminant, its default) src/textures/imagemap.h:113-117, src/core/spectrum.cpp:98-180
// The pyramid is the host's (the reference's own construction); only the per-hit lookup runs here. A texture value
// stays in the compact form of FromRGB (two basis indices, three weights) and its 31 bins are formed where the
// spectral loops need them, like every other spectrum on this path.
#pragma once
#include "d_sampling.h"

namespace dpt {

struct CamDifferentials {  // RayDifferential's offset rays (geometry.h:897-931)
    V3 rxOrigin, ryOrigin, rxDirection, ryDirection;
};

// PerspectiveCamera::GenerateRayDifferential (perspective.cpp:95-146) + ScaleDifferentials(s) (geometry.h:917-922)
// for the camera sample (pFilm, lens) whose main ray is (o, d) in world space.
DEV CamDifferentials CameraDifferentials(const DScene &s, float pFilmX, float pFilmY, float lensU, float lensV, const V3 &o, const V3 &d,
                                         float scale) {
    const mi_camera &cam = s.camera;
    const V3 pCamera = XfPoint(cam.raster_to_camera, V3(pFilmX, pFilmY, 0));
    const V3 r0 = XfPoint(cam.raster_to_camera, V3(0, 0, 0));
    const V3 dxCamera = XfPoint(cam.raster_to_camera, V3(1, 0, 0)) - r0, dyCamera = XfPoint(cam.raster_to_camera, V3(0, 1, 0)) - r0;
    V3 rxO, ryO, rxD, ryD;
    if (cam.lens_radius > 0) {
        float lx, ly;
        ConcentricSampleDisk(lensU, lensV, &lx, &ly);
        lx = cam.lens_radius * lx; ly = cam.lens_radius * ly;
        V3 dx = Normalize(pCamera + dxCamera);
        float ft = cam.focal_distance / dx.z;
        V3 pFocus = V3(0, 0, 0) + (ft * dx);
        rxO = V3(lx, ly, 0);
        rxD = Normalize(pFocus - rxO);
        V3 dy = Normalize(pCamera + dyCamera);
        ft = cam.focal_distance / dy.z;
        pFocus = V3(0, 0, 0) + (ft * dy);
        ryO = V3(lx, ly, 0);
        ryD = Normalize(pFocus - ryO);
    } else {
        rxO = ryO = V3(0, 0, 0);
        rxD = Normalize(pCamera + dxCamera);
        ryD = Normalize(pCamera + dyCamera);
    }
    CamDifferentials c;
    c.rxOrigin = XfPoint(cam.camera_to_world, rxO);
    c.ryOrigin = XfPoint(cam.camera_to_world, ryO);
    c.rxDirection = XfVector(cam.camera_to_world, rxD);
    c.ryDirection = XfVector(cam.camera_to_world, ryD);
    c.rxOrigin = o + (c.rxOrigin - o) * scale;
    c.ryOrigin = o + (c.ryOrigin - o) * scale;
    c.rxDirection = d + (c.rxDirection - d) * scale;
    c.ryDirection = d + (c.ryDirection - d) * scale;
    return c;
}

struct TexDifferentials { float dudx, dvdx, dudy, dvdy; };

DEV bool SolveLinearSystem2x2(const float A[2][2], const float B[2], float *x0, float *x1) {  // transform.cpp:41-49
    float det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    if (absf(det) < 1e-10f) return false;
    *x0 = (A[1][1] * B[0] - A[0][1] * B[1]) / det;
    *x1 = (A[0][0] * B[1] - A[1][0] * B[0]) / det;
    if (isnanf_(*x0) || isnanf_(*x1)) return false;
    return true;
}

DEV TexDifferentials ComputeDifferentials(const V3 &p, const V3 &n, const V3 &dpdu, const V3 &dpdv, const CamDifferentials &ray) {
    TexDifferentials t;
    t.dudx = t.dvdx = t.dudy = t.dvdy = 0;
    float d = Dot(n, V3(p.x, p.y, p.z));
    float tx = -(Dot(n, ray.rxOrigin) - d) / Dot(n, ray.rxDirection);
    if (isinff(tx) || isnanf_(tx)) return t;
    V3 px = ray.rxOrigin + tx * ray.rxDirection;
    float ty = -(Dot(n, ray.ryOrigin) - d) / Dot(n, ray.ryDirection);
    if (isinff(ty) || isnanf_(ty)) return t;
    V3 py = ray.ryOrigin + ty * ray.ryDirection;
    int dim0, dim1;
    if (absf(n.x) > absf(n.y) && absf(n.x) > absf(n.z)) { dim0 = 1; dim1 = 2; }
    else if (absf(n.y) > absf(n.z)) { dim0 = 0; dim1 = 2; }
    else { dim0 = 0; dim1 = 1; }
    float A[2][2] = {{dpdu[dim0], dpdv[dim0]}, {dpdu[dim1], dpdv[dim1]}};
    float Bx[2] = {px[dim0] - p[dim0], px[dim1] - p[dim1]};
    float By[2] = {py[dim0] - p[dim0], py[dim1] - p[dim1]};
    if (!SolveLinearSystem2x2(A, Bx, &t.dudx, &t.dvdx)) t.dudx = t.dvdx = 0;
    if (!SolveLinearSystem2x2(A, By, &t.dudy, &t.dvdy)) t.dudy = t.dvdy = 0;
    return t;
}

// (u, v) of a triangle hit, the geometric dpdv that goes with TriInteraction's dpdu (triangle.cpp:293-330), and the rest of
// the shading geometry Triangle::Intersect sets (triangle.cpp:347-413): shading.dpdv, shading.dndu / dndv, and whether the
// shape flips normals -- what Material::Bump reads besides what TriInteraction already returned.
struct TriShading {
    V3 dpdv;          // geometric
    V3 shDpdv, dndu, dndv;
    bool flip;
};
DEV void TriTexCoords(const DScene &s, int tri, float b0, float b1, float b2, const SurfaceInteraction &si, float *u, float *v, TriShading *ts) {
    const int32_t *vi = &s.triIndices[3 * tri];
    V3 p0 = LoadV3(s.P, vi[0]), p1 = LoadV3(s.P, vi[1]), p2 = LoadV3(s.P, vi[2]);
    const mi_mesh m = s.meshes[s.triMesh[tri]];
    float uv[3][2];
    GetUVs(s, tri, m, uv);
    V3 dpdu;
    TriPartials(p0, p1, p2, uv, &dpdu, &ts->dpdv);
    *u = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];
    *v = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
    ts->flip = (m.flags & MI_MESH_FLIP) != 0;
    ts->shDpdv = ts->dpdv;
    ts->dndu = ts->dndv = V3(0, 0, 0);
    if (m.flags & MI_MESH_HAS_N) {
        V3 n0 = LoadV3(s.N, vi[0]), n1 = LoadV3(s.N, vi[1]), n2 = LoadV3(s.N, vi[2]);
        // ts of the (ss, ts) pair: TriInteraction left ss in si.shDpdu; ss = Cross(ts, ns) there, so recompute ts as it did
        V3 ng = Normalize(Cross(p0 - p2, p1 - p2));
        V3 ns = (b0 * n0 + b1 * n1 + b2 * n2);
        if (ns.LengthSquared() > 0) ns = Normalize(ns);
        else ns = ng;
        V3 ss = Normalize(dpdu);
        V3 tsv = Cross(ss, ns);
        if (tsv.LengthSquared() > 0.f) tsv = Normalize(tsv);
        else CoordinateSystem(ns, &ss, &tsv);
        ts->shDpdv = tsv;
        float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]};
        float duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
        V3 dn1 = n0 - n2, dn2 = n1 - n2;
        float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
        bool degenerateUV = absf(determinant) < 1e-8;
        if (degenerateUV) {
            V3 dn = Cross(n2 - n0, n1 - n0);
            if (dn.LengthSquared() != 0) CoordinateSystem(dn, &ts->dndu, &ts->dndv);
        } else {
            float invDet = 1 / determinant;
            ts->dndu = (duv12[1] * dn1 - duv02[1] * dn2) * invDet;
            ts->dndv = (-duv12[0] * dn1 + duv02[0] * dn2) * invDet;
        }
    }
}

struct RGB3 {
    float r, g, b;
    DEV RGB3() : r(0), g(0), b(0) {}
    DEV RGB3(float r, float g, float b) : r(r), g(g), b(b) {}
    DEV RGB3 operator+(const RGB3 &o) const { return RGB3(r + o.r, g + o.g, b + o.b); }
    DEV RGB3 operator*(float a) const { return RGB3(r * a, g * a, b * a); }
};
DEV RGB3 operator*(float a, const RGB3 &s) { return s * a; }

DEV int MipUSize(const mi_mipmap &m, int level) { return max(1, m.width >> level); }
DEV int MipVSize(const mi_mipmap &m, int level) { return max(1, m.height >> level); }
DEV RGB3 MipTexel(const mi_mipmap &m, int level, int s, int t) {  // mipmap.h:213-235
    const int w = MipUSize(m, level), h = MipVSize(m, level);
    if (m.wrap == 0) { s = ModI(s, w); t = ModI(t, h); }
    else if (m.wrap == 2) { s = min(max(s, 0), w - 1); t = min(max(t, 0), h - 1); }
    else if (s < 0 || s >= w || t < 0 || t >= h) return RGB3();
    const float *px = m.texels + 3 * ((size_t)m.level_offset[level] + (size_t)t * w + s);
    return RGB3(px[0], px[1], px[2]);
}
DEV RGB3 MipTriangle(const mi_mipmap &m, int level, float st0, float st1) {  // mipmap.h:268-279
    level = min(max(level, 0), m.n_levels - 1);
    float s = st0 * MipUSize(m, level) - 0.5f;
    float t = st1 * MipVSize(m, level) - 0.5f;
    int s0 = (int)floorf(s), t0 = (int)floorf(t);
    float ds = s - s0, dt = t - t0;
    return (1 - ds) * (1 - dt) * MipTexel(m, level, s0, t0) + (1 - ds) * dt * MipTexel(m, level, s0, t0 + 1) +
           ds * (1 - dt) * MipTexel(m, level, s0 + 1, t0) + ds * dt * MipTexel(m, level, s0 + 1, t0 + 1);
}
DEV float Log2F(float x) { const float invLog2 = 1.442695040888963387004650940071; return logF(x) * invLog2; }
DEV RGB3 MipLookupWidth(const mi_mipmap &m, float st0, float st1, float width, bool noFiltering) {  // mipmap.h:238-266
    if (noFiltering) {
        float s = st0 * MipUSize(m, 0) - 0.5f;
        float t = st1 * MipVSize(m, 0) - 0.5f;
        return MipTexel(m, 0, (int)roundf(s), (int)roundf(t));
    }
    float level = m.n_levels - 1 + Log2F(maxf(width, 1e-8f));
    if (level < 0) return MipTriangle(m, 0, st0, st1);
    else if (level >= m.n_levels - 1) return MipTexel(m, m.n_levels - 1, 0, 0);
    int iLevel = (int)floorf(level);
    float delta = level - iLevel;
    return (1 - delta) * MipTriangle(m, iLevel, st0, st1) + delta * MipTriangle(m, iLevel + 1, st0, st1);
}
DEV RGB3 MipEWA(const DScene &s, const mi_mipmap &m, int level, float st0, float st1, float d00, float d01, float d10, float d11) {  // mipmap.h:321-380
    if (level >= m.n_levels) return MipTexel(m, m.n_levels - 1, 0, 0);
    const int us = MipUSize(m, level), vs = MipVSize(m, level);
    st0 = st0 * us - 0.5f;
    st1 = st1 * vs - 0.5f;
    d00 *= us; d01 *= vs; d10 *= us; d11 *= vs;
    float A = d01 * d01 + d11 * d11 + 1;
    float B = -2 * (d00 * d01 + d10 * d11);
    float C = d00 * d00 + d10 * d10 + 1;
    float invF = 1 / (A * C - B * B * 0.25f);
    A *= invF; B *= invF; C *= invF;
    float det = -B * B + 4 * A * C;
    float invDet = 1 / det;
    float uSqrt = __builtin_sqrtf(det * C), vSqrt = __builtin_sqrtf(A * det);
    int s0 = (int)ceilf(st0 - 2 * invDet * uSqrt);
    int s1 = (int)floorf(st0 + 2 * invDet * uSqrt);
    int t0 = (int)ceilf(st1 - 2 * invDet * vSqrt);
    int t1 = (int)floorf(st1 + 2 * invDet * vSqrt);
    RGB3 sum;
    float sumWts = 0;
    for (int it = t0; it <= t1; ++it) {
        float tt = it - st1;
        for (int is = s0; is <= s1; ++is) {
            float ss = is - st0;
            float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 < 1) {
                int index = min((int)(r2 * 128), 128 - 1);
                float weight = s.ewaWeights[index];
                sum = sum + MipTexel(m, level, is, it) * weight;
                sumWts += weight;
            }
        }
    }
    return RGB3(sum.r / sumWts, sum.g / sumWts, sum.b / sumWts);
}
DEV RGB3 MipLookup(const DScene &s, const mi_mipmap &m, float st0, float st1, float dx0, float dx1, float dy0, float dy1, int filter,
                   float maxAnisotropy) {  // mipmap.h:281-319
    if (filter != MI_TEX_EWA) {
        float width = maxf(maxf(absf(dx0), absf(dx1)), maxf(absf(dy0), absf(dy1)));
        return MipLookupWidth(m, st0, st1, 2 * width, filter == MI_TEX_NONE);
    }
    float a0 = dx0, a1 = dx1, b0 = dy0, b1 = dy1;   // dst0 = a, dst1 = b
    if (a0 * a0 + a1 * a1 < b0 * b0 + b1 * b1) { float t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; }
    float majorLength = __builtin_sqrtf(a0 * a0 + a1 * a1);
    float minorLength = __builtin_sqrtf(b0 * b0 + b1 * b1);
    if (minorLength * maxAnisotropy < majorLength && minorLength > 0) {
        float scale = majorLength / (minorLength * maxAnisotropy);
        b0 *= scale; b1 *= scale;
        minorLength *= scale;
    }
    if (minorLength == 0) return MipTriangle(m, 0, st0, st1);
    float lod = maxf(0.f, m.n_levels - 1.f + Log2F(minorLength));
    int ilod = (int)floorf(lod);
    float t = lod - ilod;
    return (1 - t) * MipEWA(s, m, ilod, st0, st1, a0, a1, b0, b1) + t * MipEWA(s, m, ilod + 1, st0, st1, a0, a1, b0, b1);
}

// The alpha tests of Triangle::Intersect / IntersectP (triangle.cpp:331-338, 531-570) for a candidate hit of triangle
// `tri` at barycentrics (b0, b1, b2): false = the hit does not count. The texture is evaluated as the reference
// evaluates it on isectLocal: (u, v) of the hit, zero footprint. `shadow`: IntersectP also tests "shadowalpha".
DEV float AlphaTextureValue(const DScene &s, int tex, float u, float v) {
    const mi_texture &t = s.textures[tex];
    const mi_mipmap &m = s.mipmaps[t.mipmap];
    return MipLookup(s, m, t.su * u + t.du, t.sv * v + t.dv, 0.f, 0.f, 0.f, 0.f, t.filter, t.max_aniso).r * t.post_scale;
}
DEV bool AlphaPass(const DScene &s, int tri, float b0, float b1, float b2, bool shadow) {
    const mi_mesh m = s.meshes[s.triMesh[tri]];
    float uv[3][2];
    GetUVs(s, tri, m, uv);
    const float u = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];
    const float v = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
    if (m.alpha_tex >= 0 && AlphaTextureValue(s, m.alpha_tex, u, v) == 0.f) return false;
    if (shadow && m.shadow_alpha_tex >= 0 && AlphaTextureValue(s, m.shadow_alpha_tex, u, v) == 0.f) return false;
    return true;
}

// The same for a sphere hit (sphere.cpp:113-160): (u, v) = (phi / phiMax, (theta - thetaMin) / (thetaMax - thetaMin)), dpdv
// and the Weingarten dndu / dndv, taken to world space as Transform::operator()(SurfaceInteraction) does.
DEV bool SphereTexCoords(const mi_sphere &sp, const V3 &ro, const V3 &rd, float *u, float *v, TriShading *ts) {
    V3 dObj, pHit; float phi, t;
    if (!SphereRoots<true>(sp, ro, rd, kInfinity, &dObj, &pHit, &phi, &t)) return false;
    const float radius = sp.radius, phiMax = sp.phi_max, thetaMin = sp.theta_min, thetaMax = sp.theta_max;
    *u = phi / phiMax;
    float theta = acosF(clampf(pHit.z / radius, -1, 1));
    *v = (theta - thetaMin) / (thetaMax - thetaMin);
    float zRadius = __builtin_sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    float invZRadius = 1 / zRadius;
    float cosPhi = pHit.x * invZRadius, sinPhi = pHit.y * invZRadius;
    V3 dpdu(-phiMax * pHit.y, phiMax * pHit.x, 0);
    V3 dpdv = (thetaMax - thetaMin) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinF(theta));
    V3 d2Pduu = -phiMax * phiMax * V3(pHit.x, pHit.y, 0);
    V3 d2Pduv = (thetaMax - thetaMin) * pHit.z * phiMax * V3(-sinPhi, cosPhi, 0.);
    V3 d2Pdvv = -(thetaMax - thetaMin) * (thetaMax - thetaMin) * V3(pHit.x, pHit.y, pHit.z);
    float E = Dot(dpdu, dpdu), F = Dot(dpdu, dpdv), G = Dot(dpdv, dpdv);
    V3 N = Normalize(Cross(dpdu, dpdv));
    float e = Dot(N, d2Pduu), f = Dot(N, d2Pduv), g = Dot(N, d2Pdvv);
    float invEGF2 = 1 / (E * G - F * F);
    V3 dndu = (f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv;
    V3 dndv = (g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv;
    ts->dpdv = XfVector(sp.o2w, dpdv);
    ts->shDpdv = ts->dpdv;
    ts->dndu = XfNormal(sp.w2o, dndu);
    ts->dndv = XfNormal(sp.w2o, dndv);
    ts->flip = (sp.reverse_orientation != 0) ^ (sp.swaps_handedness != 0);
    return true;
}

// Texture<Float>::Evaluate(si) of float image texture `tex`
DEV float EvalFloatImageTexture(const DScene &s, int tex, float u, float v, const TexDifferentials &td) {
    const mi_texture &t = s.textures[tex];
    const mi_mipmap &m = s.mipmaps[t.mipmap];
    return MipLookup(s, m, t.su * u + t.du, t.sv * v + t.dv, t.su * td.dudx, t.sv * td.dvdx, t.su * td.dudy, t.sv * td.dvdy, t.filter, t.max_aniso).r * t.post_scale;
}
// Material::Bump (material.cpp:47-84) with a uv-mapped displacement: updates the interaction's shading normal and
// shading dpdu (what the BSDF frame is built from).
DEV void Bump(const DScene &s, int tex, float u, float v, const TexDifferentials &td, const TriShading &tsh, SurfaceInteraction *si) {
    float du = .5f * (absf(td.dudx) + absf(td.dudy));
    if (du == 0) du = .0005f;
    const float uDisplace = EvalFloatImageTexture(s, tex, u + du, v, td);
    float dv = .5f * (absf(td.dvdx) + absf(td.dvdy));
    if (dv == 0) dv = .0005f;
    const float vDisplace = EvalFloatImageTexture(s, tex, u, v + dv, td);
    const float displace = EvalFloatImageTexture(s, tex, u, v, td);
    const V3 dpdu = si->shDpdu + (uDisplace - displace) / du * si->shN + displace * tsh.dndu;
    const V3 dpdv = tsh.shDpdv + (vDisplace - displace) / dv * si->shN + displace * tsh.dndv;
    V3 n = Normalize(Cross(dpdu, dpdv));   // SetShadingGeometry(..., false), interaction.cpp:76-93
    if (tsh.flip) n = -n;
    si->shN = Faceforward(n, si->n);
    si->shDpdu = dpdu;
}

// Texture<Spectrum>::Evaluate(si) of image texture `tex` in FromRGB's compact form (see IllumRGB)
// Checkerboard2DTexture::Evaluate over constant tex1 / tex2 (checkerboard.h:47-86): the weight of tex2
DEV float CheckerboardArea2(float st0, float st1, float dx0, float dx1, float dy0, float dy1, bool aaNone) {
    const float point = (((int)floorf(st0) + (int)floorf(st1)) % 2 == 0) ? 0.f : 1.f;
    if (aaNone) return point;
    const float ds = maxf(absf(dx0), absf(dy0)), dt = maxf(absf(dx1), absf(dy1));
    const float s0 = st0 - ds, s1 = st0 + ds, t0 = st1 - dt, t1 = st1 + dt;
    if (floorf(s0) == floorf(s1) && floorf(t0) == floorf(t1)) return point;
    auto bumpInt = [](float x) { return (int)floorf(x / 2) + 2 * maxf(x / 2 - (int)floorf(x / 2) - 0.5f, 0.f); };
    const float sint = (bumpInt(s1) - bumpInt(s0)) / (2 * ds);
    const float tint = (bumpInt(t1) - bumpInt(t0)) / (2 * dt);
    float area2 = sint + tint - 2 * sint * tint;
    if (ds > 1 || dt > 1) area2 = .5f;
    return area2;
}
DEV IllumRGB EvalImageTexture(const DScene &s, int tex, float u, float v, const TexDifferentials &td) {
    const mi_texture &t = s.textures[tex];
    if (t.type == MI_TEX_CHECKERBOARD) {
        IllumRGB q;
        q.i1 = -1; q.i2 = tex; q.w1 = q.w2 = 0.f;
        q.w0 = CheckerboardArea2(t.su * u + t.du, t.sv * v + t.dv, t.su * td.dudx, t.sv * td.dvdx, t.su * td.dudy, t.sv * td.dvdy, t.aa_none != 0);
        return q;
    }
    const mi_mipmap &m = s.mipmaps[t.mipmap];
    const RGB3 mem = MipLookup(s, m, t.su * u + t.du, t.sv * v + t.dv, t.su * td.dudx, t.sv * td.dvdx, t.su * td.dudy, t.sv * td.dvdy,
                               t.filter, t.max_aniso);
    const float rgb[3] = {mem.r, mem.g, mem.b};
    return MakeIllumRGB(rgb);
}
}  // namespace dpt
