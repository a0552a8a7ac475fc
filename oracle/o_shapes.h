// oracle/o_shapes.h -- TEST INFRASTRUCTURE (CPU oracle).
// Triangle / sphere intersection, surface interactions and area sampling,
// restating src/shapes/triangle.cpp:188-608, src/shapes/sphere.cpp:49-306,
// src/core/efloat.h:48-300, src/core/transform.h:222-400, src/core/shape.cpp:56-87,
// src/core/interaction.cpp:44-90 against the flat scene description.
#pragma once
#include "../include/mi_pt.h"
#include "o_math.h"

namespace orc {

struct Interaction {
    V3 p, pError, wo, n;
};

struct SurfaceInteraction : Interaction {
    Float uv[2] = {0, 0};
    V3 dpdu, dpdv, dndu, dndv;
    struct { V3 n, dpdu, dpdv, dndu, dndv; } shading;
    int prim = -1;  // index into desc.prims
    bool flip = false;  // shape->reverseOrientation ^ shape->transformSwapsHandedness
};

inline Ray SpawnRay(const Interaction &it, const V3 &d) {  // interaction.h:64-67
    return Ray(OffsetRayOrigin(it.p, it.pError, it.n, d), d, Infinity);
}
inline Ray SpawnRayTo(const Interaction &a, const Interaction &b) {  // interaction.h:73-78
    V3 origin = OffsetRayOrigin(a.p, a.pError, a.n, b.p - a.p);
    V3 target = OffsetRayOrigin(b.p, b.pError, b.n, origin - b.p);
    V3 d = target - origin;
    return Ray(origin, d, 1 - ShadowEpsilon);
}

// SurfaceInteraction ctor, interaction.cpp:44-74
inline void InitSurfaceInteraction(SurfaceInteraction *si, const V3 &p, const V3 &pError, Float u, Float v,
                                   const V3 &wo, const V3 &dpdu, const V3 &dpdv, const V3 &dndu,
                                   const V3 &dndv, bool flip) {
    si->p = p;
    si->pError = pError;
    si->wo = Normalize(wo);  // Interaction ctor, interaction.h:60
    si->n = Normalize(Cross(dpdu, dpdv));
    si->uv[0] = u; si->uv[1] = v;
    si->dpdu = dpdu; si->dpdv = dpdv; si->dndu = dndu; si->dndv = dndv;
    si->shading.n = si->n;
    si->shading.dpdu = dpdu; si->shading.dpdv = dpdv; si->shading.dndu = dndu; si->shading.dndv = dndv;
    if (flip) { si->n *= -1; si->shading.n *= -1; }
    si->flip = flip;
}

// ---------------------------------------------------------------- triangles
struct TriHit { Float t, b0, b1, b2; };

struct TriVerts {
    V3 p0, p1, p2;
};
inline TriVerts GetTri(const mi_scene_desc &d, int tri) {
    const int32_t *v = &d.tri_indices[3 * tri];
    TriVerts t;
    t.p0 = V3(d.P[3 * v[0]], d.P[3 * v[0] + 1], d.P[3 * v[0] + 2]);
    t.p1 = V3(d.P[3 * v[1]], d.P[3 * v[1] + 1], d.P[3 * v[1] + 2]);
    t.p2 = V3(d.P[3 * v[2]], d.P[3 * v[2] + 1], d.P[3 * v[2] + 2]);
    return t;
}
inline void GetUVs(const mi_scene_desc &d, int tri, Float uv[3][2]) {  // triangle.h:98-108
    const mi_mesh &m = d.meshes[d.tri_mesh[tri]];
    if (m.flags & MI_MESH_HAS_UV) {
        const int32_t *v = &d.tri_indices[3 * tri];
        for (int i = 0; i < 3; ++i) { uv[i][0] = d.UV[2 * v[i]]; uv[i][1] = d.UV[2 * v[i] + 1]; }
    } else {
        uv[0][0] = 0; uv[0][1] = 0; uv[1][0] = 1; uv[1][1] = 0; uv[2][0] = 1; uv[2][1] = 1;
    }
}

// The watertight test shared by Intersect and IntersectP, triangle.cpp:199-291 / 437-526.
inline bool TriTest(const V3 &p0, const V3 &p1, const V3 &p2, const Ray &ray, TriHit *hit) {
    V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
    int kz = MaxDimension(Abs(ray.d));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    V3 d = Permute(ray.d, kx, ky, kz);
    p0t = Permute(p0t, kx, ky, kz);
    p1t = Permute(p1t, kx, ky, kz);
    p2t = Permute(p2t, kx, ky, kz);
    Float Sx = -d.x / d.z, Sy = -d.y / d.z, Sz = 1.f / d.z;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y;
        double p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y;
        double p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y;
        double p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    Float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    Float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < ray.tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > ray.tMax * det)) return false;
    Float invDet = 1 / det;
    Float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    Float t = tScaled * invDet;
    Float maxZt = MaxComponent(Abs(V3(p0t.z, p1t.z, p2t.z)));
    Float deltaZ = gamma(3) * maxZt;
    Float maxXt = MaxComponent(Abs(V3(p0t.x, p1t.x, p2t.x)));
    Float maxYt = MaxComponent(Abs(V3(p0t.y, p1t.y, p2t.y)));
    Float deltaX = gamma(5) * (maxXt + maxZt);
    Float deltaY = gamma(5) * (maxYt + maxZt);
    Float deltaE = 2 * (gamma(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    Float maxE = MaxComponent(Abs(V3(e0, e1, e2)));
    Float deltaT = 3 * (gamma(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * std::abs(invDet);
    if (t <= deltaT) return false;
    hit->t = t; hit->b0 = b0; hit->b1 = b1; hit->b2 = b2;
    return true;
}

// Texture<Float>::Evaluate of an "alpha" texture at (u, v) with the zero footprint of Triangle::Intersect's isectLocal
// (triangle.cpp:331-338); defined in o_texture.h.
Float AlphaTextureValue(const mi_scene_desc &d, int tex, Float u, Float v);

// dpdu/dpdv for a triangle; false when the triangle itself is degenerate
// (triangle.cpp:293-317: such an intersection is rejected by Intersect).
inline bool TriPartials(const mi_scene_desc &d, int tri, const TriVerts &tv, V3 *dpdu, V3 *dpdv) {
    Float uv[3][2];
    GetUVs(d, tri, uv);
    Float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]};
    Float duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
    V3 dp02 = tv.p0 - tv.p2, dp12 = tv.p1 - tv.p2;
    Float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
    bool degenerateUV = std::abs(determinant) < 1e-8;
    if (!degenerateUV) {
        Float invdet = 1 / determinant;
        *dpdu = (duv12[1] * dp02 - duv02[1] * dp12) * invdet;
        *dpdv = (-duv12[0] * dp02 + duv02[0] * dp12) * invdet;
    }
    if (degenerateUV || Cross(*dpdu, *dpdv).LengthSquared() == 0) {
        V3 ng = Cross(tv.p2 - tv.p0, tv.p1 - tv.p0);
        if (ng.LengthSquared() == 0) return false;
        CoordinateSystem(Normalize(ng), dpdu, dpdv);
    }
    return true;
}

// Triangle::IntersectP, triangle.cpp:427-574: the t test, then -- only for meshes with an alpha mask -- the degenerate-
// triangle rejection and the "alpha" / "shadowalpha" tests.
inline bool TriIntersectP(const mi_scene_desc &d, int tri, const Ray &ray) {
    TriVerts tv = GetTri(d, tri);
    TriHit h;
    if (!TriTest(tv.p0, tv.p1, tv.p2, ray, &h)) return false;
    const mi_mesh &mesh = d.meshes[d.tri_mesh[tri]];
    if (mesh.alpha_tex >= 0 || mesh.shadow_alpha_tex >= 0) {
        V3 dpdu, dpdv;
        if (!TriPartials(d, tri, tv, &dpdu, &dpdv)) return false;
        Float uv[3][2];
        GetUVs(d, tri, uv);
        Float uHit = h.b0 * uv[0][0] + h.b1 * uv[1][0] + h.b2 * uv[2][0];
        Float vHit = h.b0 * uv[0][1] + h.b1 * uv[1][1] + h.b2 * uv[2][1];
        if (mesh.alpha_tex >= 0 && AlphaTextureValue(d, mesh.alpha_tex, uHit, vHit) == 0) return false;
        if (mesh.shadow_alpha_tex >= 0 && AlphaTextureValue(d, mesh.shadow_alpha_tex, uHit, vHit) == 0) return false;
    }
    return true;
}

// Triangle::Intersect, triangle.cpp:188-425.
inline bool TriIntersect(const mi_scene_desc &d, int tri, const Ray &ray, Float *tHit, SurfaceInteraction *isect) {
    TriVerts tv = GetTri(d, tri);
    TriHit h;
    if (!TriTest(tv.p0, tv.p1, tv.p2, ray, &h)) return false;
    V3 dpdu, dpdv;
    if (!TriPartials(d, tri, tv, &dpdu, &dpdv)) return false;
    const V3 &p0 = tv.p0, &p1 = tv.p1, &p2 = tv.p2;
    Float b0 = h.b0, b1 = h.b1, b2 = h.b2;
    Float xAbsSum = (std::abs(b0 * p0.x) + std::abs(b1 * p1.x) + std::abs(b2 * p2.x));
    Float yAbsSum = (std::abs(b0 * p0.y) + std::abs(b1 * p1.y) + std::abs(b2 * p2.y));
    Float zAbsSum = (std::abs(b0 * p0.z) + std::abs(b1 * p1.z) + std::abs(b2 * p2.z));
    V3 pError = gamma(7) * V3(xAbsSum, yAbsSum, zAbsSum);
    V3 pHit = b0 * p0 + b1 * p1 + b2 * p2;
    Float uv[3][2];
    GetUVs(d, tri, uv);
    Float uHit = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];
    Float vHit = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
    const mi_mesh &mesh = d.meshes[d.tri_mesh[tri]];
    if (mesh.alpha_tex >= 0 && AlphaTextureValue(d, mesh.alpha_tex, uHit, vHit) == 0) return false;   // triangle.cpp:331-338
    bool flip = (mesh.flags & MI_MESH_FLIP) != 0;
    InitSurfaceInteraction(isect, pHit, pError, uHit, vHit, -ray.d, dpdu, dpdv, V3(0, 0, 0), V3(0, 0, 0), flip);
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    isect->n = isect->shading.n = Normalize(Cross(dp02, dp12));
    if (mesh.flags & MI_MESH_HAS_N) {
        const int32_t *v = &d.tri_indices[3 * tri];
        V3 n0(d.N[3 * v[0]], d.N[3 * v[0] + 1], d.N[3 * v[0] + 2]);
        V3 n1(d.N[3 * v[1]], d.N[3 * v[1] + 1], d.N[3 * v[1] + 2]);
        V3 n2(d.N[3 * v[2]], d.N[3 * v[2] + 1], d.N[3 * v[2] + 2]);
        V3 ns = (b0 * n0 + b1 * n1 + b2 * n2);
        if (ns.LengthSquared() > 0) ns = Normalize(ns);
        else ns = isect->n;
        V3 ss = Normalize(isect->dpdu);  // mesh->s is not carried on this path
        V3 ts = Cross(ss, ns);
        if (ts.LengthSquared() > 0.f) {
            ts = Normalize(ts);
            ss = Cross(ts, ns);
        } else
            CoordinateSystem(ns, &ss, &ts);
        V3 dndu, dndv;
        Float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]};
        Float duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
        V3 dn1 = n0 - n2, dn2 = n1 - n2;
        Float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
        bool degenerateUV = std::abs(determinant) < 1e-8;
        if (degenerateUV) {
            V3 dn = Cross(n2 - n0, n1 - n0);
            if (dn.LengthSquared() == 0) dndu = dndv = V3(0, 0, 0);
            else CoordinateSystem(dn, &dndu, &dndv);
        } else {
            Float invDet = 1 / determinant;
            dndu = (duv12[1] * dn1 - duv02[1] * dn2) * invDet;
            dndv = (-duv12[0] * dn1 + duv02[0] * dn2) * invDet;
        }
        // SetShadingGeometry(ss, ts, dndu, dndv, true), interaction.cpp:76-93
        isect->shading.n = Normalize(Cross(ss, ts));
        if (flip) isect->shading.n = -isect->shading.n;
        isect->n = Faceforward(isect->n, isect->shading.n);
        isect->shading.dpdu = ss; isect->shading.dpdv = ts;
        isect->shading.dndu = dndu; isect->shading.dndv = dndv;
        isect->n = Faceforward(isect->n, isect->shading.n);
    } else if (flip)
        isect->n = isect->shading.n = -isect->n;
    *tHit = h.t;
    return true;
}

inline Float TriArea(const TriVerts &tv) { return 0.5 * Cross(tv.p1 - tv.p0, tv.p2 - tv.p0).Length(); }

// Triangle::Sample(u), triangle.cpp:583-608
inline Interaction TriSample(const mi_scene_desc &d, int tri, const Float u[2], Float *pdf) {
    Float su0 = std::sqrt(u[0]);
    Float b[2] = {1 - su0, u[1] * su0};  // UniformSampleTriangle
    TriVerts tv = GetTri(d, tri);
    Interaction it;
    it.p = b[0] * tv.p0 + b[1] * tv.p1 + (1 - b[0] - b[1]) * tv.p2;
    it.n = Normalize(Cross(tv.p1 - tv.p0, tv.p2 - tv.p0));
    const mi_mesh &mesh = d.meshes[d.tri_mesh[tri]];
    if (mesh.flags & MI_MESH_HAS_N) {
        const int32_t *v = &d.tri_indices[3 * tri];
        V3 n0(d.N[3 * v[0]], d.N[3 * v[0] + 1], d.N[3 * v[0] + 2]);
        V3 n1(d.N[3 * v[1]], d.N[3 * v[1] + 1], d.N[3 * v[1] + 2]);
        V3 n2(d.N[3 * v[2]], d.N[3 * v[2] + 1], d.N[3 * v[2] + 2]);
        V3 ns(b[0] * n0 + b[1] * n1 + (1 - b[0] - b[1]) * n2);
        it.n = Faceforward(it.n, ns);
    } else if (mesh.flags & MI_MESH_FLIP)
        it.n *= -1;
    V3 pAbsSum = Abs(b[0] * tv.p0) + Abs(b[1] * tv.p1) + Abs((1 - b[0] - b[1]) * tv.p2);
    it.pError = gamma(6) * V3(pAbsSum.x, pAbsSum.y, pAbsSum.z);
    *pdf = 1 / TriArea(tv);
    return it;
}

// ---------------------------------------------------------------- EFloat
struct EFloat {  // efloat.h:48-200 (NDEBUG build)
    float v, low, high;
    EFloat() {}
    EFloat(float v, float err = 0.f) : v(v) {
        if (err == 0.) low = high = v;
        else { low = NextFloatDown(v - err); high = NextFloatUp(v + err); }
    }
    EFloat operator+(EFloat ef) const {
        EFloat r; r.v = v + ef.v;
        r.low = NextFloatDown(low + ef.low); r.high = NextFloatUp(high + ef.high);
        return r;
    }
    EFloat operator-(EFloat ef) const {
        EFloat r; r.v = v - ef.v;
        r.low = NextFloatDown(low - ef.high); r.high = NextFloatUp(high - ef.low);
        return r;
    }
    EFloat operator*(EFloat ef) const {
        EFloat r; r.v = v * ef.v;
        Float prod[4] = {low * ef.low, high * ef.low, low * ef.high, high * ef.high};
        r.low = NextFloatDown(std::min(std::min(prod[0], prod[1]), std::min(prod[2], prod[3])));
        r.high = NextFloatUp(std::max(std::max(prod[0], prod[1]), std::max(prod[2], prod[3])));
        return r;
    }
    EFloat operator/(EFloat ef) const {
        EFloat r; r.v = v / ef.v;
        if (ef.low < 0 && ef.high > 0) { r.low = -Infinity; r.high = Infinity; }
        else {
            Float div[4] = {low / ef.low, high / ef.low, low / ef.high, high / ef.high};
            r.low = NextFloatDown(std::min(std::min(div[0], div[1]), std::min(div[2], div[3])));
            r.high = NextFloatUp(std::max(std::max(div[0], div[1]), std::max(div[2], div[3])));
        }
        return r;
    }
    bool operator==(EFloat fe) const { return v == fe.v; }
    float UpperBound() const { return high; }
    float LowerBound() const { return low; }
    explicit operator float() const { return v; }
};
inline EFloat operator*(float f, EFloat fe) { return EFloat(f) * fe; }
inline bool Quadratic(EFloat A, EFloat B, EFloat C, EFloat *t0, EFloat *t1) {  // efloat.h:271-290
    double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    double rootDiscrim = std::sqrt(discrim);
    EFloat floatRootDiscrim(rootDiscrim, MachineEpsilon * rootDiscrim);
    EFloat q;
    if ((float)B < 0) q = -.5 * (B - floatRootDiscrim);
    else q = -.5 * (B + floatRootDiscrim);
    *t0 = q / A;
    *t1 = C / q;
    if ((float)*t0 > (float)*t1) std::swap(*t0, *t1);
    return true;
}

// ---------------------------------------------------------------- transforms (row-major m[16])
inline V3 XfPoint(const float *m, const V3 &p) {  // transform.h:222-233
    Float x = p.x, y = p.y, z = p.z;
    Float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    Float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    Float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    Float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1) return V3(xp, yp, zp);
    Float inv = (Float)1 / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
inline V3 XfPointErr(const float *m, const V3 &p, V3 *pError) {  // transform.h:269-290
    Float x = p.x, y = p.y, z = p.z;
    Float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    Float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    Float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    Float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    Float xAbsSum = (std::abs(m[0] * x) + std::abs(m[1] * y) + std::abs(m[2] * z) + std::abs(m[3]));
    Float yAbsSum = (std::abs(m[4] * x) + std::abs(m[5] * y) + std::abs(m[6] * z) + std::abs(m[7]));
    Float zAbsSum = (std::abs(m[8] * x) + std::abs(m[9] * y) + std::abs(m[10] * z) + std::abs(m[11]));
    *pError = gamma(3) * V3(xAbsSum, yAbsSum, zAbsSum);
    if (wp == 1) return V3(xp, yp, zp);
    Float inv = (Float)1 / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
inline V3 XfPointErr2(const float *m, const V3 &pt, const V3 &ptError, V3 *absError) {  // transform.h:292-322
    Float x = pt.x, y = pt.y, z = pt.z;
    Float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    Float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    Float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    Float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    absError->x = (gamma(3) + (Float)1) * (std::abs(m[0]) * ptError.x + std::abs(m[1]) * ptError.y + std::abs(m[2]) * ptError.z) +
                  gamma(3) * (std::abs(m[0] * x) + std::abs(m[1] * y) + std::abs(m[2] * z) + std::abs(m[3]));
    absError->y = (gamma(3) + (Float)1) * (std::abs(m[4]) * ptError.x + std::abs(m[5]) * ptError.y + std::abs(m[6]) * ptError.z) +
                  gamma(3) * (std::abs(m[4] * x) + std::abs(m[5] * y) + std::abs(m[6] * z) + std::abs(m[7]));
    absError->z = (gamma(3) + (Float)1) * (std::abs(m[8]) * ptError.x + std::abs(m[9]) * ptError.y + std::abs(m[10]) * ptError.z) +
                  gamma(3) * (std::abs(m[8] * x) + std::abs(m[9] * y) + std::abs(m[10] * z) + std::abs(m[11]));
    if (wp == 1.) return V3(xp, yp, zp);
    Float inv = (Float)1 / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
inline V3 XfVector(const float *m, const V3 &v) {  // transform.h:236-241
    Float x = v.x, y = v.y, z = v.z;
    return V3(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
}
inline V3 XfVectorErr(const float *m, const V3 &v, V3 *absError) {  // transform.h:324-340
    Float x = v.x, y = v.y, z = v.z;
    absError->x = gamma(3) * (std::abs(m[0] * v.x) + std::abs(m[1] * v.y) + std::abs(m[2] * v.z));
    absError->y = gamma(3) * (std::abs(m[4] * v.x) + std::abs(m[5] * v.y) + std::abs(m[6] * v.z));
    absError->z = gamma(3) * (std::abs(m[8] * v.x) + std::abs(m[9] * v.y) + std::abs(m[10] * v.z));
    return V3(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
}
// Normal transform by a Transform whose INVERSE is mInv (transform.h:244-249)
inline V3 XfNormal(const float *mInv, const V3 &n) {
    Float x = n.x, y = n.y, z = n.z;
    return V3(mInv[0] * x + mInv[4] * y + mInv[8] * z, mInv[1] * x + mInv[5] * y + mInv[9] * z,
              mInv[2] * x + mInv[6] * y + mInv[10] * z);
}
// Transform::operator()(const Ray&, Vector3f *oError, Vector3f *dError), transform.h:372-384
inline Ray XfRayErr(const float *m, const Ray &r, V3 *oError, V3 *dError) {
    V3 o = XfPointErr(m, r.o, oError);
    V3 d = XfVectorErr(m, r.d, dError);
    Float lengthSquared = d.LengthSquared();
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(d), *oError) / lengthSquared;
        o += d * dt;
    }
    return Ray(o, d, r.tMax);
}
// Transform::operator()(const Ray&), transform.h:251-266 (camera rays)
inline Ray XfRay(const float *m, const Ray &r) {
    V3 oError;
    V3 o = XfPointErr(m, r.o, &oError);
    V3 d = XfVector(m, r.d);
    Float lengthSquared = d.LengthSquared();
    Float tMax = r.tMax;
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(d), oError) / lengthSquared;
        o += d * dt;
        tMax -= dt;
    }
    return Ray(o, d, tMax);
}

// ---------------------------------------------------------------- spheres
// Shared quadric root selection of Sphere::Intersect / IntersectP, sphere.cpp:49-112,158-214.
inline bool SphereRoots(const mi_sphere &s, const Ray &r, Ray *rayObj, V3 *pHitOut, Float *phiOut, Float *tOut) {
    V3 oErr, dErr;
    Ray ray = XfRayErr(s.w2o, r, &oErr, &dErr);
    EFloat ox(ray.o.x, oErr.x), oy(ray.o.y, oErr.y), oz(ray.o.z, oErr.z);
    EFloat dx(ray.d.x, dErr.x), dy(ray.d.y, dErr.y), dz(ray.d.z, dErr.z);
    EFloat a = dx * dx + dy * dy + dz * dz;
    EFloat b = 2 * (dx * ox + dy * oy + dz * oz);
    EFloat c = ox * ox + oy * oy + oz * oz - EFloat(s.radius) * EFloat(s.radius);
    EFloat t0, t1;
    if (!Quadratic(a, b, c, &t0, &t1)) return false;
    if (t0.UpperBound() > ray.tMax || t1.LowerBound() <= 0) return false;
    EFloat tShapeHit = t0;
    if (tShapeHit.LowerBound() <= 0) {
        tShapeHit = t1;
        if (tShapeHit.UpperBound() > ray.tMax) return false;
    }
    const Float radius = s.radius, zMin = s.z_min, zMax = s.z_max, phiMax = s.phi_max;
    V3 pHit = ray((Float)tShapeHit);
    pHit *= radius / Distance(pHit, V3(0, 0, 0));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
    Float phi = Atan2F(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * Pi;
    if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) {
        if (tShapeHit == t1) return false;
        if (t1.UpperBound() > ray.tMax) return false;
        tShapeHit = t1;
        pHit = ray((Float)tShapeHit);
        pHit *= radius / Distance(pHit, V3(0, 0, 0));
        if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
        phi = Atan2F(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * Pi;
        if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) return false;
    }
    *rayObj = ray; *pHitOut = pHit; *phiOut = phi; *tOut = (Float)tShapeHit;
    return true;
}
inline bool SphereIntersectP(const mi_sphere &s, const Ray &r) {
    Ray ro; V3 pHit; Float phi, t;
    return SphereRoots(s, r, &ro, &pHit, &phi, &t);
}
inline bool SphereIntersect(const mi_sphere &s, const Ray &r, Float *tHit, SurfaceInteraction *isect) {
    Ray ray; V3 pHit; Float phi, t;
    if (!SphereRoots(s, r, &ray, &pHit, &phi, &t)) return false;
    const Float radius = s.radius, phiMax = s.phi_max, thetaMin = s.theta_min, thetaMax = s.theta_max;
    Float u = phi / phiMax;
    Float theta = AcosF(Clamp(pHit.z / radius, -1, 1));
    Float v = (theta - thetaMin) / (thetaMax - thetaMin);
    Float zRadius = std::sqrt(pHit.x * pHit.x + pHit.y * pHit.y);
    Float invZRadius = 1 / zRadius;
    Float cosPhi = pHit.x * invZRadius;
    Float sinPhi = pHit.y * invZRadius;
    V3 dpdu(-phiMax * pHit.y, phiMax * pHit.x, 0);
    V3 dpdv = (thetaMax - thetaMin) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * SinF(theta));
    V3 d2Pduu = -phiMax * phiMax * V3(pHit.x, pHit.y, 0);
    V3 d2Pduv = (thetaMax - thetaMin) * pHit.z * phiMax * V3(-sinPhi, cosPhi, 0.);
    V3 d2Pdvv = -(thetaMax - thetaMin) * (thetaMax - thetaMin) * V3(pHit.x, pHit.y, pHit.z);
    Float E = Dot(dpdu, dpdu), F = Dot(dpdu, dpdv), G = Dot(dpdv, dpdv);
    V3 N = Normalize(Cross(dpdu, dpdv));
    Float e = Dot(N, d2Pduu), f = Dot(N, d2Pduv), g = Dot(N, d2Pdvv);
    Float invEGF2 = 1 / (E * G - F * F);
    V3 dndu = (f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv;
    V3 dndv = (g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv;
    V3 pError = gamma(5) * Abs(pHit);
    SurfaceInteraction o;
    bool flip = (s.reverse_orientation != 0) ^ (s.swaps_handedness != 0);
    InitSurfaceInteraction(&o, pHit, pError, u, v, -ray.d, dpdu, dpdv, dndu, dndv, flip);
    // (*ObjectToWorld)(SurfaceInteraction), transform.cpp:255-288
    const float *m = s.o2w, *mi = s.w2o;
    isect->p = XfPointErr2(m, o.p, o.pError, &isect->pError);
    isect->n = Normalize(XfNormal(mi, o.n));
    isect->wo = Normalize(XfVector(m, o.wo));
    isect->uv[0] = o.uv[0]; isect->uv[1] = o.uv[1];
    isect->dpdu = XfVector(m, o.dpdu); isect->dpdv = XfVector(m, o.dpdv);
    isect->dndu = XfNormal(mi, o.dndu); isect->dndv = XfNormal(mi, o.dndv);
    isect->shading.n = Normalize(XfNormal(mi, o.shading.n));
    isect->shading.dpdu = XfVector(m, o.shading.dpdu); isect->shading.dpdv = XfVector(m, o.shading.dpdv);
    isect->shading.dndu = XfNormal(mi, o.shading.dndu); isect->shading.dndv = XfNormal(mi, o.shading.dndv);
    isect->shading.n = Faceforward(isect->shading.n, isect->n);
    isect->flip = flip;
    *tHit = t;
    return true;
}
inline Float SphereArea(const mi_sphere &s) { return s.phi_max * s.radius * (s.z_max - s.z_min); }

inline V3 UniformSampleSphere(const Float u[2]) {  // sampling.cpp:98-103
    Float z = 1 - 2 * u[0];
    Float r = std::sqrt(std::max((Float)0, (Float)1 - z * z));
    Float phi = 2 * Pi * u[1];
    return V3(r * CosF(phi), r * SinF(phi), z);
}
// Sphere::Sample(u), sphere.cpp:219-230
inline Interaction SphereSampleArea(const mi_sphere &s, const Float u[2], Float *pdf) {
    V3 pObj = V3(0, 0, 0) + s.radius * UniformSampleSphere(u);
    Interaction it;
    it.n = Normalize(XfNormal(s.w2o, V3(pObj.x, pObj.y, pObj.z)));
    if (s.reverse_orientation) it.n *= -1;
    pObj *= s.radius / Distance(pObj, V3(0, 0, 0));
    V3 pObjError = gamma(5) * Abs(pObj);
    it.p = XfPointErr2(s.o2w, pObj, pObjError, &it.pError);
    *pdf = 1 / SphereArea(s);
    return it;
}
// Sphere::Sample(ref,u), sphere.cpp:232-292
inline Interaction SphereSample(const mi_sphere &s, const Interaction &ref, const Float u[2], Float *pdf) {
    V3 pCenter = XfPoint(s.o2w, V3(0, 0, 0));
    const Float radius = s.radius;
    V3 pOrigin = OffsetRayOrigin(ref.p, ref.pError, ref.n, pCenter - ref.p);
    if (DistanceSquared(pOrigin, pCenter) <= radius * radius) {
        Interaction intr = SphereSampleArea(s, u, pdf);
        V3 wi = intr.p - ref.p;
        if (wi.LengthSquared() == 0) *pdf = 0;
        else {
            wi = Normalize(wi);
            *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
        }
        if (std::isinf(*pdf)) *pdf = 0.f;
        return intr;
    }
    V3 wc = Normalize(pCenter - ref.p);
    V3 wcX, wcY;
    CoordinateSystem(wc, &wcX, &wcY);
    Float sinThetaMax2 = radius * radius / DistanceSquared(ref.p, pCenter);
    Float cosThetaMax = std::sqrt(std::max((Float)0, 1 - sinThetaMax2));
    Float cosTheta = (1 - u[0]) + u[0] * cosThetaMax;
    Float sinTheta = std::sqrt(std::max((Float)0, 1 - cosTheta * cosTheta));
    Float phi = u[1] * 2 * Pi;
    Float dc = Distance(ref.p, pCenter);
    Float ds = dc * cosTheta - std::sqrt(std::max((Float)0, radius * radius - dc * dc * sinTheta * sinTheta));
    Float cosAlpha = (dc * dc + radius * radius - ds * ds) / (2 * dc * radius);
    Float sinAlpha = std::sqrt(std::max((Float)0, 1 - cosAlpha * cosAlpha));
    V3 nWorld = SphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc);
    V3 pWorld = pCenter + radius * V3(nWorld.x, nWorld.y, nWorld.z);
    Interaction it;
    it.p = pWorld;
    it.pError = gamma(5) * Abs(pWorld);
    it.n = nWorld;
    if (s.reverse_orientation) it.n *= -1;
    *pdf = 1 / (2 * Pi * (1 - cosThetaMax));
    return it;
}

}  // namespace orc
