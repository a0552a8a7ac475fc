// d_scene.h -- device-resident scene (HBM layout) and the shape routines the
// wavefront kernels call: watertight ray/triangle test, sphere quadric with
// interval arithmetic, surface-interaction construction and area sampling.
//
// HBM layout (all read-only after mi_pt_create):
//   nodes    32 B/node = LinearBVHNode bytes (bvh.cpp:95-104), kept for reference/debug
//   wnodes   64 B per INTERIOR node, four 16-B loads: the boxes of both children
//            {Lmin.xyz,Lmax.x} {Lmax.yz,Rmin.xy} {Rmin.z,Rmax.xyz} and
//            {childL, childR, nPrimsL|axis<<16, nPrimsR}; child = interior index, or first
//            primitive when nPrims > 0. One fetch yields both child tests, halving the
//            dependent-load chain of a ray; leaves are never fetched as nodes.
//   primTri  48 B/primitive in BVH leaf order: p0|flags, p1|shape, p2|0  (positions are
//            pre-gathered so a leaf test is three coalescable 16-B loads, no index chase)
//   the indexed mesh (tri_indices, P, N, UV), spheres, materials, lights as in mi_pt.h
// Algorithms restate src/shapes/triangle.cpp:188-608, src/shapes/sphere.cpp:49-306,
// src/core/efloat.h, src/core/transform.h:222-400, src/core/interaction.cpp:44-93.
#pragma once
#include "../../../include/mi_pt.h"
#include "d_math.h"

namespace dpt {

#define PRIM_FLAG_SPHERE 1u
#define PRIM_FLAG_DEGENERATE 2u
#define PRIM_FLAG_INSTANCE 64u   // a TransformedPrimitive: primTri[3 * prim + 1].w holds the instance number (mi_prim.instance - 1)
#define PRIM_FLAG_ALPHA 32u   // the triangle's mesh has an "alpha" / "shadowalpha" mask (mi_mesh.alpha_tex)
#define PRIM_CLASS_SHIFT 8      /* bits 8-11: shading class of the primitive's material (15 = no BSDF) */

struct DScene {
    const float4 *nodes;
    const float4 *wnodes;  // wide nodes (see pt_kernels.hip): bvhWidth 2 -> 64 B per BVH2 interior node (both children's boxes);
                           // bvhWidth 4 -> 128 B per two-level subtree (up to four grandchild boxes + the visit orders)
    int bvhWidth;
    // ObjectInstance as TransformedPrimitive (mi_instance): the instances, the wide record each one's tree starts at, and
    // that tree's root box (two float4 per instance: min, max)
    const mi_instance *instances;
    const int32_t *instWideRoot;
    const float4 *instRootBounds;
    uint32_t nInstances;
    const float4 *primTri;
    const mi_prim *prims;
    const int32_t *triIndices;
    const uint32_t *triMesh;
    const float *P, *N, *UV;
    const mi_mesh *meshes;
    const mi_sphere *spheres;
    const mi_material *materials;
    const mi_light *lights;
    const int *lightPrim;        // [nLights]: the primitive whose shape an area light is (k_trav<3> leaves it out)
    int misAny;                  // MIS rays are traced as visibility queries (k_trav, MODE 3; pt_kernels.hip)
    const float4 *lightBounds;   // [2 * nLights]: dilated world bounds of an area light's shape (see F_MIS_DARK, pt_kernels.hip)
    uint32_t nNodes, nPrims, nLights, nMaterials;
    uint32_t classMask;  // shading classes present in the scene (bit c), see pt_kernels.hip
    // light distribution: distribution d at func[d*nLights], cdf[d*(nLights+1)], funcInt[d]
    int ldType;
    int nVoxels[3];
    const float *ldFunc, *ldCdf, *ldFuncInt;
    float wbMin[3], wbMax[3];
    // sampler tables
    const int32_t *primes, *primeSums;
    const uint16_t *perms;
    const uint64_t *primeMagic;        // ceil(2^64 / prime): a / prime == umul64hi(a, magic) for a < 2^32
    const uint32_t *pixelOffsetTable;  // 128x128 Halton per-pixel index offsets (halton.cpp:98-118)
    // image textures (ABI v5): device copies of mi_texture / mi_mipmap (texel pointers in HBM),
    // MIPMap::weightLut (mipmap.h:199-206, tabulated by the host), 1 / sqrt(samplesPerPixel) for ScaleDifferentials
    const mi_texture *textures;
    const mi_mipmap *mipmaps;
    const float *ewaWeights;
    float invSqrtSpp;
    const float *filterTable;  // 256 floats
    float cieY[MI_NSPEC];
    const mi_envmap *envmaps;      // device copies: the pointers inside point to device memory
    const float *rgbIllum;         // [7][31] rgbIllum2Spect White, Cyan, Magenta, Yellow, Red, Green, Blue
    int infiniteLights[4];         // indices of the MI_LIGHT_INFINITE lights (scene.infiniteLights), -1 = none
    int nInfiniteLights;
    mi_camera camera;
    // film
    int croppedBounds[4], sampleBounds[4], pixelBounds[4];
    float filterRadius[2];
    float maxSampleLuminance;
    // sampler
    int baseScales[2], baseExponents[2], sampleStride, multInverse[2], sampleAtPixelCenter;
    int samplerType;                       // mi_sampler_type
    int sobolResolution, sobolLog2Resolution;
    const uint32_t *sobolMatrices;         // [n_sobol_dims * 52]
    const uint64_t *sobolVdc, *sobolVdcInv;
    long long samplesPerPixel;             // (RANDOM: the stream number of a camera sample)
    // pixel samplers (ZEROTWO / STRATIFIED): the tables of every pixel of the sample bounds, built at create (k_pixel_tables):
    // pixTab1[(pixel * pixelDims + dimension) * spp + sample], pixTab2[((pixel * pixelDims + dimension) * spp + sample) * 2 + {0, 1}]
    // set per render (mi_pt_render): the Halton indices of the pass fit 32 bits (I_IDXHI is neither stored nor read); somebody
    // reads a path's pixel and sample number after k_generate (other samplers, spectralpath bands, textured lens cameras)
    int index32, storePixelSample;
    const float *pixTab1, *pixTab2;
    int pixelDims, xSamples, ySamples, jitter;
    // integrator
    int maxDepth;
    float rrThreshold;
    int nBands, bandDelta;  // spectralpath: paths per camera sample, bins per band (1, 31 for "path")
};

struct Interaction {
    V3 p, pError, wo, n;
};
struct SurfaceInteraction : Interaction {
    V3 dpdu;       // geometric dpdu (world)
    V3 shN;        // shading.n
    V3 shDpdu;     // shading.dpdu
};

DEV Ray SpawnRay(const Interaction &it, const V3 &d) { return Ray(OffsetRayOrigin(it.p, it.pError, it.n, d), d, kInfinity); }
DEV Ray SpawnRayTo(const Interaction &a, const Interaction &b) {  // interaction.h:73-78
    V3 origin = OffsetRayOrigin(a.p, a.pError, a.n, b.p - a.p);
    V3 target = OffsetRayOrigin(b.p, b.pError, b.n, origin - b.p);
    V3 d = target - origin;
    return Ray(origin, d, 1 - kShadowEpsilon);
}

// ------------------------------------------------------------------ triangles
struct TriHit { float t, b0, b1, b2; };

// triangle.cpp:199-291 / 437-526. The permutation and the shear depend on the ray alone:
// TriRay holds them (same expressions, evaluated once per ray instead of once per test).
struct TriRay {
    int kz;
    float Sx, Sy, Sz;
};
DEV TriRay MakeTriRay(const V3 &rd) {
    TriRay tr;
    tr.kz = MaxDimension(Abs(rd));
    int kx = tr.kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    V3 d = V3(rd[kx], rd[ky], rd[tr.kz]);
    tr.Sx = -d.x / d.z; tr.Sy = -d.y / d.z; tr.Sz = 1.f / d.z;
    return tr;
}
DEV V3 PermuteZ(const V3 &v, int kz) {  // (v[kx], v[ky], v[kz]) with kx = kz+1, ky = kz+2 (mod 3)
    // written as scalar selects so that it compiles to v_cndmask instead of branches
    const bool k0 = kz == 0, k1 = kz == 1;
    const float x = k0 ? v.y : (k1 ? v.z : v.x);
    const float y = k0 ? v.z : (k1 ? v.x : v.y);
    const float z = k0 ? v.x : (k1 ? v.y : v.z);
    return V3(x, y, z);
}
// (detOut / tScaledOut: the two quantities of the only tMax-dependent line of the test, for callers that re-apply it with a
// smaller tMax -- the cooperative leaf test of k_trav)
DEV bool TriTestRay(const V3 &p0, const V3 &p1, const V3 &p2, const V3 &ro, const TriRay &tr, float tMax, TriHit *hit,
                    float *detOut = nullptr, float *tScaledOut = nullptr) {
    V3 p0t = PermuteZ(p0 - ro, tr.kz), p1t = PermuteZ(p1 - ro, tr.kz), p2t = PermuteZ(p2 - ro, tr.kz);
    const float Sx = tr.Sx, Sy = tr.Sy, Sz = tr.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
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
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = MaxComponent(Abs(V3(p0t.z, p1t.z, p2t.z)));
    float deltaZ = gammaf(3) * maxZt;
    float maxXt = MaxComponent(Abs(V3(p0t.x, p1t.x, p2t.x)));
    float maxYt = MaxComponent(Abs(V3(p0t.y, p1t.y, p2t.y)));
    float deltaX = gammaf(5) * (maxXt + maxZt);
    float deltaY = gammaf(5) * (maxYt + maxZt);
    float deltaE = 2 * (gammaf(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = MaxComponent(Abs(V3(e0, e1, e2)));
    float deltaT = 3 * (gammaf(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * absf(invDet);
    if (t <= deltaT) return false;
    hit->t = t; hit->b0 = b0; hit->b1 = b1; hit->b2 = b2;
    if (detOut) { *detOut = det; *tScaledOut = tScaled; }
    return true;
}

DEV bool TriTest(const V3 &p0, const V3 &p1, const V3 &p2, const V3 &ro, const V3 &rd, float tMax, TriHit *hit) {
    return TriTestRay(p0, p1, p2, ro, MakeTriRay(rd), tMax, hit);
}

DEV V3 LoadV3(const float *a, int i) { return V3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }

DEV void GetUVs(const DScene &s, int tri, const mi_mesh &m, float uv[3][2]) {
    if (m.flags & MI_MESH_HAS_UV) {
        const int32_t *v = &s.triIndices[3 * tri];
        for (int i = 0; i < 3; ++i) { uv[i][0] = s.UV[2 * v[i]]; uv[i][1] = s.UV[2 * v[i] + 1]; }
    } else {
        uv[0][0] = 0; uv[0][1] = 0; uv[1][0] = 1; uv[1][1] = 0; uv[2][0] = 1; uv[2][1] = 1;
    }
}

// dpdu/dpdv (triangle.cpp:293-317). Returns false for a degenerate triangle.
DEV bool TriPartials(const V3 &p0, const V3 &p1, const V3 &p2, const float uv[3][2], V3 *dpdu, V3 *dpdv) {
    float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]};
    float duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
    bool degenerateUV = absf(determinant) < 1e-8;
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        *dpdu = (duv12[1] * dp02 - duv02[1] * dp12) * invdet;
        *dpdv = (-duv12[0] * dp02 + duv02[0] * dp12) * invdet;
    }
    if (degenerateUV || Cross(*dpdu, *dpdv).LengthSquared() == 0) {
        V3 ng = Cross(p2 - p0, p1 - p0);
        if (ng.LengthSquared() == 0) return false;
        CoordinateSystem(Normalize(ng), dpdu, dpdv);
    }
    return true;
}

// SurfaceInteraction of a triangle hit from (tri, barycentrics), the part of
// Triangle::Intersect after the t test (triangle.cpp:319-423), reduced to what the
// PathIntegrator consumes with constant textures: p, pError, n, wo, shading.n,
// shading.dpdu, dpdu.
DEV void TriInteraction(const DScene &s, int tri, float b0, float b1, float b2, const V3 &rayD, SurfaceInteraction *si) {
    const int32_t *v = &s.triIndices[3 * tri];
    V3 p0 = LoadV3(s.P, v[0]), p1 = LoadV3(s.P, v[1]), p2 = LoadV3(s.P, v[2]);
    const mi_mesh m = s.meshes[s.triMesh[tri]];
    float uv[3][2];
    GetUVs(s, tri, m, uv);
    V3 dpdu, dpdv;
    TriPartials(p0, p1, p2, uv, &dpdu, &dpdv);
    float xAbsSum = (absf(b0 * p0.x) + absf(b1 * p1.x) + absf(b2 * p2.x));
    float yAbsSum = (absf(b0 * p0.y) + absf(b1 * p1.y) + absf(b2 * p2.y));
    float zAbsSum = (absf(b0 * p0.z) + absf(b1 * p1.z) + absf(b2 * p2.z));
    si->pError = gammaf(7) * V3(xAbsSum, yAbsSum, zAbsSum);
    si->p = b0 * p0 + b1 * p1 + b2 * p2;
    si->wo = Normalize(-rayD);
    si->dpdu = dpdu;
    bool flip = (m.flags & MI_MESH_FLIP) != 0;
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    V3 n = Normalize(Cross(dp02, dp12));
    V3 shN = n;
    V3 shDpdu = dpdu;
    if (m.flags & MI_MESH_HAS_N) {
        V3 n0 = LoadV3(s.N, v[0]), n1 = LoadV3(s.N, v[1]), n2 = LoadV3(s.N, v[2]);
        V3 ns = (b0 * n0 + b1 * n1 + b2 * n2);
        if (ns.LengthSquared() > 0) ns = Normalize(ns);
        else ns = n;
        V3 ss = Normalize(dpdu);
        V3 ts = Cross(ss, ns);
        if (ts.LengthSquared() > 0.f) {
            ts = Normalize(ts);
            ss = Cross(ts, ns);
        } else
            CoordinateSystem(ns, &ss, &ts);
        // SetShadingGeometry(ss, ts, ..., true), interaction.cpp:76-93
        shN = Normalize(Cross(ss, ts));
        if (flip) shN = -shN;
        n = Faceforward(n, shN);
        shDpdu = ss;
        n = Faceforward(n, shN);
    } else if (flip) {
        n = -n;
        shN = n;
    }
    si->n = n;
    si->shN = shN;
    si->shDpdu = shDpdu;
}

DEV float TriAreaOf(const V3 &p0, const V3 &p1, const V3 &p2) { return (float)(0.5 * (double)Cross(p1 - p0, p2 - p0).Length()); }

// Triangle::Sample(u), triangle.cpp:583-608
DEV Interaction TriSample(const DScene &s, int tri, float u0, float u1, float *pdf) {
    float su0 = __builtin_sqrtf(u0);
    float b[2] = {1 - su0, u1 * su0};
    const int32_t *v = &s.triIndices[3 * tri];
    V3 p0 = LoadV3(s.P, v[0]), p1 = LoadV3(s.P, v[1]), p2 = LoadV3(s.P, v[2]);
    Interaction it;
    it.p = b[0] * p0 + b[1] * p1 + (1 - b[0] - b[1]) * p2;
    it.n = Normalize(Cross(p1 - p0, p2 - p0));
    const mi_mesh m = s.meshes[s.triMesh[tri]];
    if (m.flags & MI_MESH_HAS_N) {
        V3 n0 = LoadV3(s.N, v[0]), n1 = LoadV3(s.N, v[1]), n2 = LoadV3(s.N, v[2]);
        V3 ns(b[0] * n0 + b[1] * n1 + (1 - b[0] - b[1]) * n2);
        it.n = Faceforward(it.n, ns);
    } else if (m.flags & MI_MESH_FLIP)
        it.n *= -1;
    V3 pAbsSum = Abs(b[0] * p0) + Abs(b[1] * p1) + Abs((1 - b[0] - b[1]) * p2);
    it.pError = gammaf(6) * V3(pAbsSum.x, pAbsSum.y, pAbsSum.z);
    *pdf = 1 / TriAreaOf(p0, p1, p2);
    return it;
}

// ------------------------------------------------------------------ EFloat (efloat.h:48-200)
struct EFloat {
    float v, low, high;
    DEV EFloat() {}
    DEV EFloat(float v_, float err = 0.f) : v(v_) {
        if (err == 0.) low = high = v_;
        else { low = NextFloatDown(v_ - err); high = NextFloatUp(v_ + err); }
    }
    DEV EFloat operator+(EFloat ef) const {
        EFloat r; r.v = v + ef.v;
        r.low = NextFloatDown(low + ef.low); r.high = NextFloatUp(high + ef.high);
        return r;
    }
    DEV EFloat operator-(EFloat ef) const {
        EFloat r; r.v = v - ef.v;
        r.low = NextFloatDown(low - ef.high); r.high = NextFloatUp(high - ef.low);
        return r;
    }
    DEV EFloat operator*(EFloat ef) const {
        EFloat r; r.v = v * ef.v;
        float p0 = low * ef.low, p1 = high * ef.low, p2 = low * ef.high, p3 = high * ef.high;
        r.low = NextFloatDown(minf(minf(p0, p1), minf(p2, p3)));
        r.high = NextFloatUp(maxf(maxf(p0, p1), maxf(p2, p3)));
        return r;
    }
    DEV EFloat operator/(EFloat ef) const {
        EFloat r; r.v = v / ef.v;
        if (ef.low < 0 && ef.high > 0) { r.low = -kInfinity; r.high = kInfinity; }
        else {
            float d0 = low / ef.low, d1 = high / ef.low, d2 = low / ef.high, d3 = high / ef.high;
            r.low = NextFloatDown(minf(minf(d0, d1), minf(d2, d3)));
            r.high = NextFloatUp(maxf(maxf(d0, d1), maxf(d2, d3)));
        }
        return r;
    }
};
DEV bool Quadratic(EFloat A, EFloat B, EFloat C, EFloat *t0, EFloat *t1) {  // efloat.h:271-290
    double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    double rootDiscrim = __builtin_sqrt(discrim);
    EFloat floatRootDiscrim((float)rootDiscrim, (float)((double)kMachineEpsilon * rootDiscrim));
    EFloat q;
    if (B.v < 0) q = EFloat(-.5f) * (B - floatRootDiscrim);
    else q = EFloat(-.5f) * (B + floatRootDiscrim);
    *t0 = q / A;
    *t1 = C / q;
    if (t0->v > t1->v) { EFloat tmp = *t0; *t0 = *t1; *t1 = tmp; }
    return true;
}

// ------------------------------------------------------------------ transforms (row-major m[16])
DEV V3 XfPoint(const float *m, const V3 &p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1) return V3(xp, yp, zp);
    float inv = 1.f / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
DEV V3 XfPointErr(const float *m, const V3 &p, V3 *pError) {
    float x = p.x, y = p.y, z = p.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    float xAbsSum = (absf(m[0] * x) + absf(m[1] * y) + absf(m[2] * z) + absf(m[3]));
    float yAbsSum = (absf(m[4] * x) + absf(m[5] * y) + absf(m[6] * z) + absf(m[7]));
    float zAbsSum = (absf(m[8] * x) + absf(m[9] * y) + absf(m[10] * z) + absf(m[11]));
    *pError = gammaf(3) * V3(xAbsSum, yAbsSum, zAbsSum);
    if (wp == 1) return V3(xp, yp, zp);
    float inv = 1.f / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
DEV V3 XfPointErr2(const float *m, const V3 &pt, const V3 &ptError, V3 *absError) {
    float x = pt.x, y = pt.y, z = pt.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    absError->x = (gammaf(3) + 1.f) * (absf(m[0]) * ptError.x + absf(m[1]) * ptError.y + absf(m[2]) * ptError.z) +
                  gammaf(3) * (absf(m[0] * x) + absf(m[1] * y) + absf(m[2] * z) + absf(m[3]));
    absError->y = (gammaf(3) + 1.f) * (absf(m[4]) * ptError.x + absf(m[5]) * ptError.y + absf(m[6]) * ptError.z) +
                  gammaf(3) * (absf(m[4] * x) + absf(m[5] * y) + absf(m[6] * z) + absf(m[7]));
    absError->z = (gammaf(3) + 1.f) * (absf(m[8]) * ptError.x + absf(m[9]) * ptError.y + absf(m[10]) * ptError.z) +
                  gammaf(3) * (absf(m[8] * x) + absf(m[9] * y) + absf(m[10] * z) + absf(m[11]));
    if (wp == 1.f) return V3(xp, yp, zp);
    float inv = 1.f / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
DEV V3 XfVector(const float *m, const V3 &v) {
    float x = v.x, y = v.y, z = v.z;
    return V3(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
}
DEV V3 XfVectorErr(const float *m, const V3 &v, V3 *absError) {
    float x = v.x, y = v.y, z = v.z;
    absError->x = gammaf(3) * (absf(m[0] * v.x) + absf(m[1] * v.y) + absf(m[2] * v.z));
    absError->y = gammaf(3) * (absf(m[4] * v.x) + absf(m[5] * v.y) + absf(m[6] * v.z));
    absError->z = gammaf(3) * (absf(m[8] * v.x) + absf(m[9] * v.y) + absf(m[10] * v.z));
    return V3(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
}
DEV V3 XfNormal(const float *mInv, const V3 &n) {
    float x = n.x, y = n.y, z = n.z;
    return V3(mInv[0] * x + mInv[4] * y + mInv[8] * z, mInv[1] * x + mInv[5] * y + mInv[9] * z,
              mInv[2] * x + mInv[6] * y + mInv[10] * z);
}
// Transform::operator()(const Ray&), transform.h:251-266 (camera rays)
DEV Ray XfRay(const float *m, const Ray &r) {
    V3 oError;
    V3 o = XfPointErr(m, r.o, &oError);
    V3 d = XfVector(m, r.d);
    float lengthSquared = d.LengthSquared();
    float tMax = r.tMax;
    if (lengthSquared > 0) {
        float dt = Dot(Abs(d), oError) / lengthSquared;
        o += d * dt;
        tMax -= dt;
    }
    return Ray(o, d, tMax);
}

// ------------------------------------------------------------------ spheres
// Root selection shared by Sphere::Intersect / IntersectP (sphere.cpp:49-112,158-214).
// WANT_POINT = false (SphereHitT: only t is asked for): a sphere that is not clipped -- zMin <= -radius, zMax >= radius and
// phiMax >= 2 pi AS FLOATS, so that `phi > phiMax` cannot hold for any phi that atan2 + 2 pi can produce -- passes the clipping
// tests of sphere.cpp:91-110 whatever the hit point is, so the point, its re-projection (a square root and a division) and
// phi (atan2) are not formed: a third of the function.
template <bool WANT_POINT>
__device__ __attribute__((noinline)) bool SphereRoots(const mi_sphere &s, const V3 &ro, const V3 &rd, float tMaxIn, V3 *rayObjD, V3 *pHitOut, float *phiOut, float *tOut) {
    V3 oErr, dErr;
    V3 o = XfPointErr(s.w2o, ro, &oErr);
    V3 d = XfVectorErr(s.w2o, rd, &dErr);
    {   // Transform::operator()(Ray, oErr, dErr), transform.h:372-384
        float lengthSquared = d.LengthSquared();
        if (lengthSquared > 0) {
            float dt = Dot(Abs(d), oErr) / lengthSquared;
            o += d * dt;
        }
    }
    const float rayTMax = tMaxIn;
    {   // The misses that need no error bounds. An EFloat's value is the plain float result of the same operations, and its
        // interval contains it: a negative discriminant (efloat.h:275), a nearer root whose VALUE lies beyond tMax
        // (t0.UpperBound() >= t0) and a farther root whose value is <= 0 (t1.LowerBound() <= t1) are sphere.cpp:77's misses
        // whatever the bounds are. A shadow ray towards a sphere light ends ShadowEpsilon short of the sampled point, so its
        // nearer root is ~1 against tMax = 0.9999: every NEE shadow ray of the killeroo and Cornell frames leaves here,
        // after ~40 instructions instead of the ~700 of the interval arithmetic.
        const float av = d.x * d.x + d.y * d.y + d.z * d.z;
        const float bv = 2.f * (d.x * o.x + d.y * o.y + d.z * o.z);
        const float cv = o.x * o.x + o.y * o.y + o.z * o.z - s.radius * s.radius;
        const double discrim = (double)bv * (double)bv - 4. * (double)av * (double)cv;
        if (discrim < 0.) return false;
        const float frd = (float)__builtin_sqrt(discrim);
        const float qv = (bv < 0) ? -.5f * (bv - frd) : -.5f * (bv + frd);
        float t0v = qv / av, t1v = cv / qv;
        if (t0v > t1v) { const float tmp = t0v; t0v = t1v; t1v = tmp; }
        if (t0v > rayTMax || t1v <= 0) return false;   // (a NaN compares false and goes on to the intervals)
    }
    EFloat ox(o.x, oErr.x), oy(o.y, oErr.y), oz(o.z, oErr.z);
    EFloat dx(d.x, dErr.x), dy(d.y, dErr.y), dz(d.z, dErr.z);
    EFloat a = dx * dx + dy * dy + dz * dz;
    EFloat b = EFloat(2.f) * (dx * ox + dy * oy + dz * oz);
    EFloat c = ox * ox + oy * oy + oz * oz - EFloat(s.radius) * EFloat(s.radius);
    EFloat t0, t1;
    if (!Quadratic(a, b, c, &t0, &t1)) return false;
    if (t0.high > rayTMax || t1.low <= 0) return false;
    EFloat tShapeHit = t0;
    if (tShapeHit.low <= 0) {
        tShapeHit = t1;
        if (tShapeHit.high > rayTMax) return false;
    }
    const float radius = s.radius, zMin = s.z_min, zMax = s.z_max, phiMax = s.phi_max;
    if constexpr (!WANT_POINT) {
        if (!(zMin > -radius) && !(zMax < radius) && phiMax >= 2 * kPi) { *tOut = tShapeHit.v; return true; }
    }
    V3 pHit = o + d * tShapeHit.v;
    pHit *= radius / Distance(pHit, V3(0, 0, 0));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
    float phi = atan2F(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * kPi;
    if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) {
        if (tShapeHit.v == t1.v) return false;
        if (t1.high > rayTMax) return false;
        tShapeHit = t1;
        pHit = o + d * tShapeHit.v;
        pHit *= radius / Distance(pHit, V3(0, 0, 0));
        if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
        phi = atan2F(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * kPi;
        if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) return false;
    }
    *rayObjD = d; *pHitOut = pHit; *phiOut = phi; *tOut = tShapeHit.v;
    return true;
}
DEV bool SphereHitT(const mi_sphere &s, const V3 &ro, const V3 &rd, float tMax, float *t) {
    V3 dObj, pHit; float phi;
    return SphereRoots<false>(s, ro, rd, tMax, &dObj, &pHit, &phi, t);
}
// Full Sphere::Intersect interaction (sphere.cpp:113-155 + transform.cpp:255-288); the
// ray must be the one that produced the hit (tMax = value before the hit was recorded).
DEV bool SphereInteraction(const mi_sphere &s, const V3 &ro, const V3 &rd, float tMax, SurfaceInteraction *si, float *tHit) {
    V3 dObj, pHit; float phi, t;
    if (!SphereRoots<true>(s, ro, rd, tMax, &dObj, &pHit, &phi, &t)) return false;
    const float radius = s.radius, phiMax = s.phi_max, thetaMin = s.theta_min, thetaMax = s.theta_max;
    float theta = acosF(clampf(pHit.z / radius, -1, 1));
    float zRadius = __builtin_sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    float invZRadius = 1 / zRadius;
    float cosPhi = pHit.x * invZRadius;
    float sinPhi = pHit.y * invZRadius;
    V3 dpdu(-phiMax * pHit.y, phiMax * pHit.x, 0);
    V3 dpdv = (thetaMax - thetaMin) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinF(theta));
    V3 pError = gammaf(5) * Abs(pHit);
    bool flip = (s.reverse_orientation != 0) ^ (s.swaps_handedness != 0);
    V3 nObj = Normalize(Cross(dpdu, dpdv));
    V3 shNObj = nObj;
    if (flip) { nObj *= -1; shNObj *= -1; }
    V3 woObj = Normalize(-dObj);
    const float *m = s.o2w, *mi = s.w2o;
    si->p = XfPointErr2(m, pHit, pError, &si->pError);
    si->n = Normalize(XfNormal(mi, nObj));
    si->wo = Normalize(XfVector(m, woObj));
    si->dpdu = XfVector(m, dpdu);
    si->shN = Normalize(XfNormal(mi, shNObj));
    si->shDpdu = XfVector(m, dpdu);
    si->shN = Faceforward(si->shN, si->n);
    *tHit = t;
    return true;
}
DEV float SphereArea(const mi_sphere &s) { return s.phi_max * s.radius * (s.z_max - s.z_min); }

DEV V3 UniformSampleSphere(float u0, float u1) {  // sampling.cpp:98-103
    float z = 1 - 2 * u0;
    float r = __builtin_sqrtf(maxf(0.f, 1.f - z * z));
    float phi = 2 * kPi * u1;
    return V3(r * cosF(phi), r * sinF(phi), z);
}
DEV Interaction SphereSampleArea(const mi_sphere &s, float u0, float u1, float *pdf) {  // sphere.cpp:219-230
    V3 pObj = V3(0, 0, 0) + s.radius * UniformSampleSphere(u0, u1);
    Interaction it;
    it.n = Normalize(XfNormal(s.w2o, V3(pObj.x, pObj.y, pObj.z)));
    if (s.reverse_orientation) it.n *= -1;
    pObj *= s.radius / Distance(pObj, V3(0, 0, 0));
    V3 pObjError = gammaf(5) * Abs(pObj);
    it.p = XfPointErr2(s.o2w, pObj, pObjError, &it.pError);
    *pdf = 1 / SphereArea(s);
    return it;
}
DEV Interaction SphereSample(const mi_sphere &s, const Interaction &ref, float u0, float u1, float *pdf) {  // sphere.cpp:232-292
    V3 pCenter = XfPoint(s.o2w, V3(0, 0, 0));
    const float radius = s.radius;
    V3 pOrigin = OffsetRayOrigin(ref.p, ref.pError, ref.n, pCenter - ref.p);
    if (DistanceSquared(pOrigin, pCenter) <= radius * radius) {
        Interaction intr = SphereSampleArea(s, u0, u1, pdf);
        V3 wi = intr.p - ref.p;
        if (wi.LengthSquared() == 0) *pdf = 0;
        else {
            wi = Normalize(wi);
            *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
        }
        if (isinff(*pdf)) *pdf = 0.f;
        return intr;
    }
    V3 wc = Normalize(pCenter - ref.p);
    V3 wcX, wcY;
    CoordinateSystem(wc, &wcX, &wcY);
    float sinThetaMax2 = radius * radius / DistanceSquared(ref.p, pCenter);
    float cosThetaMax = __builtin_sqrtf(maxf(0.f, 1 - sinThetaMax2));
    float cosTheta = (1 - u0) + u0 * cosThetaMax;
    float sinTheta = __builtin_sqrtf(maxf(0.f, 1 - cosTheta * cosTheta));
    float phi = u1 * 2 * kPi;
    float dc = Distance(ref.p, pCenter);
    float ds = dc * cosTheta - __builtin_sqrtf(maxf(0.f, radius * radius - dc * dc * sinTheta * sinTheta));
    float cosAlpha = (dc * dc + radius * radius - ds * ds) / (2 * dc * radius);
    float sinAlpha = __builtin_sqrtf(maxf(0.f, 1 - cosAlpha * cosAlpha));
    V3 nWorld = SphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc);
    V3 pWorld = pCenter + radius * V3(nWorld.x, nWorld.y, nWorld.z);
    Interaction it;
    it.p = pWorld;
    it.pError = gammaf(5) * Abs(pWorld);
    it.n = nWorld;
    if (s.reverse_orientation) it.n *= -1;
    *pdf = 1 / (2 * kPi * (1 - cosThetaMax));
    return it;
}

}  // namespace dpt
