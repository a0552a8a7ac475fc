// d_sampling.h -- device Halton sampler, light sampling and light-selection pmfs.
//   HaltonSampler / GlobalSampler   src/samplers/halton.cpp:98-127, src/core/sampler.cpp:46-52,136-195
//   (Scrambled)RadicalInverse       src/core/lowdiscrepancy.cpp:130-175,389-424; lowdiscrepancy.h:82-91
//   DiffuseAreaLight / Point / Distant  src/lights/diffuse.cpp:68-87, point.cpp:44-53, distant.cpp:49-59
//   Shape::Sample(ref) / Pdf(ref,wi)    src/core/shape.cpp:56-87 (+ sphere.cpp:232-306)
//   Distribution1D / SpatialLightDistribution  src/core/sampling.h:55-109, src/core/lightdistrib.cpp:135-300
#pragma once
#include "d_bsdf.h"

namespace dpt {

DEV uint32_t ReverseBits32(uint32_t n) { return __brev(n); }
DEV uint64_t ReverseBits64(uint64_t n) {
    uint64_t n0 = ReverseBits32((uint32_t)n);
    uint64_t n1 = ReverseBits32((uint32_t)(n >> 32));
    return (n0 << 32) | n1;
}

// Digit loops of the plain and scrambled radical inverse. Indices on this path fit 32
// bits (spp * 31104 < 2^32 up to 138k spp); the quotient by the (runtime) prime base is
// then one 64-bit high multiply with the precomputed magic ceil(2^64/base), exact for
// every 32-bit dividend. Wider indices take the plain 64-bit division first.
DEV uint32_t DivMagic(uint32_t a, uint64_t magic) { return (uint32_t)__umul64hi((uint64_t)a, magic); }

DEV float RadicalInverseBase(int base, uint64_t magic, uint64_t a) {
    const float invBase = 1.f / (float)base;
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    while (a >> 32) {
        uint64_t next = a / (uint64_t)base;
        uint64_t digit = a - next * (uint64_t)base;
        reversedDigits = reversedDigits * (uint64_t)base + digit;
        invBaseN *= invBase;
        a = next;
    }
    uint32_t a32 = (uint32_t)a;
    const uint32_t ub = (uint32_t)base;
    while (a32) {
        uint32_t next = DivMagic(a32, magic);
        uint32_t digit = a32 - next * ub;
        reversedDigits = reversedDigits * ub + digit;
        invBaseN *= invBase;
        a32 = next;
    }
    return minf((float)reversedDigits * invBaseN, kOneMinusEpsilon);
}
DEV float ScrambledRadicalInverseBase(int base, uint64_t magic, const uint16_t *perm, uint64_t a) {
    const float invBase = 1.f / (float)base;
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    while (a >> 32) {
        uint64_t next = a / (uint64_t)base;
        uint64_t digit = a - next * (uint64_t)base;
        reversedDigits = reversedDigits * (uint64_t)base + perm[digit];
        invBaseN *= invBase;
        a = next;
    }
    uint32_t a32 = (uint32_t)a;
    const uint32_t ub = (uint32_t)base;
    while (a32) {
        uint32_t next = DivMagic(a32, magic);
        uint32_t digit = a32 - next * ub;
        reversedDigits = reversedDigits * ub + perm[digit];
        invBaseN *= invBase;
        a32 = next;
    }
    return minf(invBaseN * ((float)reversedDigits + invBase * (float)perm[0] / (1 - invBase)), kOneMinusEpsilon);
}
DEV float RadicalInverse(const DScene &s, int baseIndex, uint64_t a) {
    if (baseIndex == 0) return (float)((double)ReverseBits64(a) * 0x1p-64);
    return RadicalInverseBase(s.primes[baseIndex], s.primeMagic[baseIndex], a);
}
DEV uint64_t InverseRadicalInverse(uint32_t base, uint64_t inverse, int nDigits) {
    uint64_t index = 0;
    for (int i = 0; i < nDigits; ++i) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}
DEV int ModI(int a, int b) { int r = a - (a / b) * b; return (r < 0) ? r + b : r; }

DEV uint64_t HaltonPixelOffset(const DScene &s, int px, int py) {  // halton.cpp:98-118, tabulated at create
    if (s.sampleStride <= 1) return 0;
    return s.pixelOffsetTable[ModI(py, 128) * 128 + ModI(px, 128)];
}
// ---- Sampler "sobol": SobolIntervalToIndex / SobolSampleFloat, lowdiscrepancy.h:229-274; sobol.cpp:42-59
static DEV_CALL float SobolSampleFloat(const uint32_t *__restrict__ matrices, uint64_t a, int dimension) {
    uint32_t v = 0;
    for (int i = dimension * MI_SOBOL_MATRIX_SIZE; a != 0; a >>= 1, i++)
        if (a & 1) v ^= matrices[i];
    return minf((float)v * 0x1p-32f, kOneMinusEpsilon);
}
// ---- Sampler "random": RNG (PCG32), rng.h:61-144; one stream per camera sample (include/mi_pt.h, mi_sampler_type)
DEV uint32_t PcgNext(uint64_t &state, uint64_t inc) {
    const uint64_t oldstate = state;
    state = oldstate * 0x5851f42d4c957f2dULL + inc;
    const uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    const uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
DEV float PcgFloat(uint64_t &state, uint64_t inc) { return minf(kOneMinusEpsilon, (float)PcgNext(state, inc) * 0x1p-32f); }
DEV uint64_t RandomStreamInc(const DScene &s, int px, int py, long long sampleNum) {
    // stream number = sample number * pixel count + pixel index: distinct for every (pixel, sample number) whatever range of
    // sample numbers a pass renders (mi_render_params.sample_begin / spp_override may go beyond the scene's samples per pixel)
    const long long w = s.sampleBounds[2] - s.sampleBounds[0], h = s.sampleBounds[3] - s.sampleBounds[1];
    const long long pix = (long long)(py - s.sampleBounds[1]) * w + (px - s.sampleBounds[0]);
    return (((uint64_t)(sampleNum * (w * h) + pix)) << 1u) | 1u;
}
// ---- the pixel samplers (mi_sampler_type: ZEROTWO, STRATIFIED): PixelSampler::Get1D / Get2D, sampler.cpp:119-135
DEV bool IsPixelSampler(const DScene &s) { return s.samplerType >= MI_SAMPLER_ZEROTWO; }
DEV long long SamplePixelCount(const DScene &s) { return (long long)(s.sampleBounds[2] - s.sampleBounds[0]) * (s.sampleBounds[3] - s.sampleBounds[1]); }
DEV long long SamplePixelIndex(const DScene &s, int px, int py) {
    return (long long)(py - s.sampleBounds[1]) * (s.sampleBounds[2] - s.sampleBounds[0]) + (px - s.sampleBounds[0]);
}
// the stream of a sample's draws beyond the tables: (sample number + 1) * pixel count + pixel index (the pixel's own stream,
// number = pixel index, makes the tables)
DEV uint64_t PixelSampleStreamInc(const DScene &s, int px, int py, long long sampleNum) {
    return (((uint64_t)((sampleNum + 1) * SamplePixelCount(s) + SamplePixelIndex(s, px, py))) << 1u) | 1u;
}
DEV uint64_t RandomStreamStart(uint64_t inc) {   // RNG::SetSequence, rng.h:98-105
    uint64_t state = 0u;
    PcgNext(state, inc);
    state += 0x853c49e6748fea9bULL;
    PcgNext(state, inc);
    return state;
}

DEV float SampleDimension(const DScene &s, uint64_t index, int dim) {  // halton.cpp:120-127 (dimensions 0 and 1 need the pixel: CameraSampleDims)
    if (s.sampleAtPixelCenter && (dim == 0 || dim == 1)) return 0.5f;
    if (dim == 0) return RadicalInverse(s, 0, index >> s.baseExponents[0]);
    else if (dim == 1) return RadicalInverse(s, 1, index / (uint64_t)s.baseScales[1]);
    else return ScrambledRadicalInverseBase(s.primes[dim], s.primeMagic[dim], &s.perms[s.primeSums[dim]], index);
}
// The same for dim >= 2 (every dimension after the film position) as one shared, out-of-line copy: the
// integrator draws 8 dimensions per path vertex, and the shading kernels are bound by instruction fetch.
static DEV_CALL float ScrambledDimension(const int32_t *primes, const int32_t *primeSums, const uint16_t *perms,
                                         const uint64_t *primeMagic, uint64_t index, int dim) {
    return ScrambledRadicalInverseBase(primes[dim], primeMagic[dim], &perms[primeSums[dim]], index);
}
DEV float SampleDimensionFrom2(const DScene &s, uint64_t index, int dim) {
    if (s.samplerType == MI_SAMPLER_SOBOL) return SobolSampleFloat(s.sobolMatrices, index, dim);
    return ScrambledDimension(s.primes, s.primeSums, s.perms, s.primeMagic, index, dim);
}

// N consecutive dimensions (>= 2) of ONE sample index, their digit loops side by side: a path vertex draws five (light
// choice, light sample, BSDF sample) and then two (next direction) dimensions of the same index, and one after the other
// every digit of every dimension waits for its own permutation lookup (~32 dependent lookups for five dimensions; in step,
// the seven of the longest chain). Per dimension the operations and their order are ScrambledRadicalInverseBase's, so the
// values are the same bits.
template <int N>
DEV void ScrambledDimensionsFused(const int32_t *__restrict__ primes, const int32_t *__restrict__ primeSums, const uint16_t *__restrict__ perms,
                                  const uint64_t *__restrict__ primeMagic, uint64_t index, int dim, float (&u)[N]) {
    if (index >> 32) {   // (beyond 138k spp: the 64-bit head of the digit loop, one dimension at a time)
#pragma unroll
        for (int k = 0; k < N; ++k) u[k] = ScrambledDimension(primes, primeSums, perms, primeMagic, index, dim + k);
        return;
    }
    uint32_t a[N], base[N];
    uint64_t magic[N], rev[N];
    float invBase[N], invBaseN[N], perm0[N];
    const uint16_t *perm[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        base[k] = (uint32_t)primes[dim + k];
        magic[k] = primeMagic[dim + k];
        perm[k] = perms + primeSums[dim + k];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        perm0[k] = (float)perm[k][0];
        invBase[k] = 1.f / (float)(int)base[k];
        invBaseN[k] = 1;
        rev[k] = 0;
        a[k] = (uint32_t)index;
    }
    while (true) {
        bool any = false;
        uint32_t next[N], digit[N];
        unsigned pv[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            next[k] = DivMagic(a[k], magic[k]);
            digit[k] = a[k] - next[k] * base[k];
            pv[k] = perm[k][digit[k]];   // (a chain that has run out reads perm[0]: unused)
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (a[k]) {
                rev[k] = rev[k] * base[k] + pv[k];
                invBaseN[k] *= invBase[k];
                a[k] = next[k];
                any = true;
            }
        if (!any) break;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) u[k] = minf(invBaseN[k] * ((float)rev[k] + invBase[k] * perm0[k] / (1 - invBase[k])), kOneMinusEpsilon);
}

// A path's view of its sampler: the sample's index (HALTON / SOBOL) or the state of its PCG32 stream (RANDOM), and the next
// dimension. Get1D = Sampler::Get1D for dimensions >= 5 (sampler.cpp:178-195, random.cpp:44-48). The stream's increment
// is recomputed from the slot's pixel and sample number when a RANDOM draw happens (`pixelWord`, `sampleNum`: planes
// I_PIXEL / I_SAMPLE), so the index-based samplers keep no extra value alive across the shading kernel.
struct PathSampler {
    uint64_t index;
    int dim;
};
// (Nothing here may take the address of the kernel's DScene argument: that would move the whole argument block into
// private memory and send every later field access through scratch -- measured: k_shade 2.3x slower.)
// HALTON_ONLY: the shading instances of Halton-sampled scenes (every BASELINE workload) are compiled without the other two
// samplers (TM_SAMPLERS, d_bsdf.h)
// A pixel sampler's state in PathSampler: `index` = the state of the sample's fall-back stream, `dim` = the next 1D table
// (bits 0-15) and the next 2D table (bits 16-31).
DEV float PixelGet1D(const DScene &s, PathSampler &ps, int px, int py, long long sampleNum) {
    const int d1 = ps.dim & 0xffff;
    if (d1 < s.pixelDims) {
        ps.dim += 1;
        return s.pixTab1[((size_t)SamplePixelIndex(s, px, py) * s.pixelDims + d1) * (size_t)s.samplesPerPixel + (size_t)sampleNum];
    }
    uint64_t state = ps.index;
    const float u = PcgFloat(state, PixelSampleStreamInc(s, px, py, sampleNum));
    ps.index = state;
    return u;
}
DEV void PixelGet2D(const DScene &s, PathSampler &ps, int px, int py, long long sampleNum, float *u0, float *u1) {
    const int d2 = (ps.dim >> 16) & 0xffff;
    if (d2 < s.pixelDims) {
        ps.dim += 0x10000;
        const float2 v = *reinterpret_cast<const float2 *>(s.pixTab2 + 2 * (((size_t)SamplePixelIndex(s, px, py) * s.pixelDims + d2) * (size_t)s.samplesPerPixel + (size_t)sampleNum));
        *u0 = v.x; *u1 = v.y;
        return;
    }
    uint64_t state = ps.index;
    const uint64_t inc = PixelSampleStreamInc(s, px, py, sampleNum);
    *u0 = PcgFloat(state, inc);
    *u1 = PcgFloat(state, inc);
    ps.index = state;
}
template <bool HALTON_ONLY = false>
DEV float Get1D(const DScene &s, PathSampler &ps, const int *__restrict__ pixelPlane, const int *__restrict__ samplePlane, uint32_t slot) {
#ifdef MIPT_EXP_FASTRNG
    if constexpr (HALTON_ONLY) {   // (timing experiment: a hash instead of the scrambled radical inverse)
        uint32_t h = (uint32_t)ps.index * 2654435761u + (uint32_t)(ps.dim++) * 0x9E3779B9u;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        return minf((float)h * 0x1p-32f, kOneMinusEpsilon);
    }
#endif
    if constexpr (HALTON_ONLY) return ScrambledDimension(s.primes, s.primeSums, s.perms, s.primeMagic, ps.index, ps.dim++);
    if (IsPixelSampler(s)) {
        const int pixelWord = pixelPlane[slot];
        return PixelGet1D(s, ps, (int)(short)(pixelWord & 0xffff), pixelWord >> 16, (long long)samplePlane[slot]);
    }
    if (s.samplerType == MI_SAMPLER_RANDOM) {
        const int pixelWord = pixelPlane[slot];
        const uint64_t inc = RandomStreamInc(s, (int)(short)(pixelWord & 0xffff), pixelWord >> 16, (long long)samplePlane[slot]);
        uint64_t state = ps.index;
        const float u = PcgFloat(state, inc);
        ps.index = state;
        ++ps.dim;
        return u;
    }
    return SampleDimensionFrom2(s, ps.index, ps.dim++);
}
// Sampler::Get2D: two consecutive dimensions of the index-based samplers and of the random stream, the next 2D table of a
// pixel sampler.
template <bool HALTON_ONLY = false>
DEV void Get2D(const DScene &s, PathSampler &ps, const int *__restrict__ pixelPlane, const int *__restrict__ samplePlane, uint32_t slot, float *u0, float *u1) {
    if constexpr (!HALTON_ONLY) {
        if (IsPixelSampler(s)) {
            const int pixelWord = pixelPlane[slot];
            PixelGet2D(s, ps, (int)(short)(pixelWord & 0xffff), pixelWord >> 16, (long long)samplePlane[slot], u0, u1);
            return;
        }
    }
    *u0 = Get1D<HALTON_ONLY>(s, ps, pixelPlane, samplePlane, slot);
    *u1 = Get1D<HALTON_ONLY>(s, ps, pixelPlane, samplePlane, slot);
}
// GetCameraSample (sampler.cpp:46-52) of sample `sampleNum` of pixel (px, py): pFilm offsets = dims 0, 1; time = dim 2;
// pLens = dims 3, 4. Returns the sample's index (or, RANDOM, its stream state after the five draws).
struct SobolCamera { uint64_t index; float u0, u1, lu, lv; };
static DEV_CALL SobolCamera CameraSampleSobol(const uint32_t *__restrict__ matrices, const uint64_t *__restrict__ vdc, const uint64_t *__restrict__ vdcInv,
                                              int log2Resolution, int resolution, int sb0, int sb1, int px, int py, uint64_t frame, bool lens) {
    SobolCamera r;
    r.lu = r.lv = 0.f;
    uint64_t index = 0;
    const uint32_t m = (uint32_t)log2Resolution;
    if (m != 0) {   // SobolIntervalToIndex, lowdiscrepancy.h:229-249
        index = frame << (m << 1);
        uint64_t delta = 0;
        for (int c = 0; frame; frame >>= 1, ++c)
            if (frame & 1) delta ^= vdc[c];
        uint64_t b = ((((uint64_t)(uint32_t)(px - sb0)) << m) | (uint64_t)(uint32_t)(py - sb1)) ^ delta;
        for (int c = 0; b; b >>= 1, ++c)
            if (b & 1) index ^= vdcInv[c];
    }
    r.index = index;
    float v0 = SobolSampleFloat(matrices, index, 0), v1 = SobolSampleFloat(matrices, index, 1);
    v0 = v0 * (float)resolution + (float)sb0;   // remap the dimensions used for the pixel sample, sobol.cpp:54-57
    v1 = v1 * (float)resolution + (float)sb1;
    r.u0 = clampf(v0 - (float)px, 0.f, kOneMinusEpsilon);
    r.u1 = clampf(v1 - (float)py, 0.f, kOneMinusEpsilon);
    if (lens) { r.lu = SobolSampleFloat(matrices, index, 3); r.lv = SobolSampleFloat(matrices, index, 4); }
    return r;
}
// *dimAfter: the path's sampler dimension after the camera sample (5; a pixel sampler's packed table counters).
DEV uint64_t CameraSampleDims(const DScene &s, int px, int py, long long sampleNum, float *u0, float *u1, float *lu, float *lv, int *dimAfter = nullptr) {
    *lu = *lv = 0.f;
    if (dimAfter) *dimAfter = 5;
    if (IsPixelSampler(s)) {   // pFilm = Get2D(), time = Get1D(), pLens = Get2D()
        PathSampler ps;
        ps.index = RandomStreamStart(PixelSampleStreamInc(s, px, py, sampleNum));
        ps.dim = 0;
        PixelGet2D(s, ps, px, py, sampleNum, u0, u1);
        (void)PixelGet1D(s, ps, px, py, sampleNum);
        PixelGet2D(s, ps, px, py, sampleNum, lu, lv);
        if (dimAfter) *dimAfter = ps.dim;
        return ps.index;
    }
    if (s.samplerType == MI_SAMPLER_RANDOM) {
        const uint64_t inc = RandomStreamInc(s, px, py, sampleNum);
        uint64_t state = RandomStreamStart(inc);
        *u0 = PcgFloat(state, inc); *u1 = PcgFloat(state, inc);
        (void)PcgFloat(state, inc);   // time
        *lu = PcgFloat(state, inc); *lv = PcgFloat(state, inc);
        return state;
    }
    if (s.samplerType == MI_SAMPLER_SOBOL) {
        const SobolCamera r = CameraSampleSobol(s.sobolMatrices, s.sobolVdc, s.sobolVdcInv, s.sobolLog2Resolution, s.sobolResolution,
                                                s.sampleBounds[0], s.sampleBounds[1], px, py, (uint64_t)sampleNum, s.camera.lens_radius > 0);
        *u0 = r.u0; *u1 = r.u1; *lu = r.lu; *lv = r.lv;
        return r.index;
    }
    const uint64_t index = HaltonPixelOffset(s, px, py) + (uint64_t)sampleNum * (uint64_t)s.sampleStride;
    *u0 = SampleDimension(s, index, 0); *u1 = SampleDimension(s, index, 1);
    if (s.camera.lens_radius > 0) { *lu = SampleDimension(s, index, 3); *lv = SampleDimension(s, index, 4); }
    return index;
}

// ------------------------------------------------------------------ lights
DEV float SpecYBinAccum(const DScene &s, int bin, float v) { return s.cieY[bin] * v; }
DEV float YScale(float yy) {  // tail of SampledSpectrum::y(), spectrum.h:418-420
    yy = (yy < 0) ? 0 : yy;
    return yy * (float)(705 - 395) / (float)(106.856895f * 31);
}
DEV float LightY(const DScene &s, const mi_light &l) {
    float yy = 0.f;
    for (int i = 0; i < MI_NSPEC; ++i) yy += s.cieY[i] * l.L[i];
    return YScale(yy);
}

DEV Interaction ShapeSample(const DScene &s, int shape, const Interaction &ref, float u0, float u1, float *pdf) {
    if (shape < 0) return SphereSample(s.spheres[~shape], ref, u0, u1, pdf);
    Interaction intr = TriSample(s, shape, u0, u1, pdf);
    V3 wi = intr.p - ref.p;
    if (wi.LengthSquared() == 0) *pdf = 0;
    else {
        wi = Normalize(wi);
        *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
        if (isinff(*pdf)) *pdf = 0.f;
    }
    return intr;
}

// Shape::Pdf(ref, wi) for the light's own shape (no BVH: it intersects only that shape).
DEV float ShapePdf(const DScene &s, int shape, float area, const Interaction &ref, const V3 &wi, float *tTri = nullptr) {
    Ray ray = SpawnRay(ref, wi);
    if (shape < 0) {
        const mi_sphere &sp = s.spheres[~shape];
        V3 pCenter = XfPoint(sp.o2w, V3(0, 0, 0));
        V3 pOrigin = OffsetRayOrigin(ref.p, ref.pError, ref.n, pCenter - ref.p);
        if (!(DistanceSquared(pOrigin, pCenter) <= sp.radius * sp.radius)) {
            float sinThetaMax2 = sp.radius * sp.radius / DistanceSquared(ref.p, pCenter);
            float cosThetaMax = __builtin_sqrtf(maxf(0.f, 1 - sinThetaMax2));
            return 1 / (2 * kPi * (1 - cosThetaMax));
        }
        SurfaceInteraction il;
        float tHit;
        if (!SphereInteraction(sp, ray.o, ray.d, kInfinity, &il, &tHit)) return 0;
        float pdf = DistanceSquared(ref.p, il.p) / (AbsDot(il.n, -wi) * area);
        if (isinff(pdf)) pdf = 0.f;
        return pdf;
    }
    const int32_t *v = &s.triIndices[3 * shape];
    V3 p0 = LoadV3(s.P, v[0]), p1 = LoadV3(s.P, v[1]), p2 = LoadV3(s.P, v[2]);
    TriHit h;
    if (!TriTest(p0, p1, p2, ray.o, ray.d, kInfinity, &h)) return 0;
    {   // Triangle::Intersect rejects degenerate triangles after the t test
        const mi_mesh m = s.meshes[s.triMesh[shape]];
        float uv[3][2];
        GetUVs(s, shape, m, uv);
        V3 dpdu, dpdv;
        if (!TriPartials(p0, p1, p2, uv, &dpdu, &dpdv)) return 0;
    }
    if (tTri) *tTri = h.t;
    SurfaceInteraction il;
    TriInteraction(s, shape, h.b0, h.b1, h.b2, ray.d, &il);
    float pdf = DistanceSquared(ref.p, il.p) / (AbsDot(il.n, -wi) * area);
    if (isinff(pdf)) pdf = 0.f;
    return pdf;
}

DEV bool IsDeltaLight(const mi_light &l) { return l.type == MI_LIGHT_POINT || l.type == MI_LIGHT_DISTANT || l.type == MI_LIGHT_SPOT; }

// ---- InfiniteAreaLight (src/lights/infinite.cpp:85-141); IllumRGB / MakeIllumRGB: d_bsdf.h
DEV float IllumBin(const DScene &s, const IllumRGB &q, int bin) {
    float r = 0.f;
    r += s.rgbIllum[bin] * q.w0;
    r += s.rgbIllum[q.i1 * MI_NSPEC + bin] * q.w1;
    r += s.rgbIllum[q.i2 * MI_NSPEC + bin] * q.w2;
    r *= .86445f;
    return clampf(r, 0.f, kInfinity);
}
// Lmap->Lookup(st), width 0: bilinear on level 0 with ImageWrap::Repeat (mipmap.h:252-281)
DEV void EnvLookup(const mi_envmap &e, float s0f, float t0f, float rgb[3]) {
    float sx = s0f * e.width - 0.5f;
    float ty = t0f * e.height - 0.5f;
    int s0 = (int)floorf(sx), t0 = (int)floorf(ty);
    float ds = sx - s0, dt = ty - t0;
    const int sA = ModI(s0, e.width), sB = ModI(s0 + 1, e.width), tA = ModI(t0, e.height), tB = ModI(t0 + 1, e.height);
    const float *p00 = &e.rgb[3 * ((size_t)tA * e.width + sA)], *p01 = &e.rgb[3 * ((size_t)tB * e.width + sA)];
    const float *p10 = &e.rgb[3 * ((size_t)tA * e.width + sB)], *p11 = &e.rgb[3 * ((size_t)tB * e.width + sB)];
    for (int k = 0; k < 3; ++k)
        rgb[k] = (1 - ds) * (1 - dt) * p00[k] + (1 - ds) * dt * p01[k] + ds * (1 - dt) * p10[k] + ds * dt * p11[k];
}
DEV V3 Mul3(const float m[9], const V3 &v) {
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
DEV float SphericalTheta(const V3 &v) { return acosF(clampf(v.z, -1, 1)); }
DEV float SphericalPhi(const V3 &v) { float p = atan2F(v.y, v.x); return (p < 0) ? (p + 2 * kPi) : p; }
constexpr float kInv2Pi = 0.15915494309189533577f;
DEV IllumRGB InfiniteLe(const DScene &s, const mi_light &l, const V3 &dir) {  // infinite.cpp:91-95
    V3 w = Normalize(Mul3(l.w2l, dir));
    float rgb[3];
    EnvLookup(s.envmaps[l.envmap], SphericalPhi(w) * kInv2Pi, SphericalTheta(w) * kInvPi, rgb);
    return MakeIllumRGB(rgb);
}
DEV float SampleContinuous1D(const float *func, const float *cdf, float funcInt, int n, float u, float *pdf, int *off) {
    int size = n + 1, first = 0, len = size;   // sampling.h:71-89
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = first - 1;
    offset = offset < 0 ? 0 : (offset > size - 2 ? size - 2 : offset);
    if (off) *off = offset;
    float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    if (pdf) *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / n;
}
DEV float InfinitePdfLi(const DScene &s, const mi_light &l, const V3 &w) {  // infinite.cpp:133-141, sampling.h:137-143
    const mi_envmap &e = s.envmaps[l.envmap];
    V3 wi = Mul3(l.w2l, w);
    float theta = SphericalTheta(wi), phi = SphericalPhi(wi);
    float sinTheta = sinF(theta);
    if (sinTheta == 0) return 0;
    int iu = (int)(phi * kInv2Pi * e.nu), iv = (int)(theta * kInvPi * e.nv);
    iu = iu < 0 ? 0 : (iu > e.nu - 1 ? e.nu - 1 : iu);
    iv = iv < 0 ? 0 : (iv > e.nv - 1 ? e.nv - 1 : iv);
    return e.cond_func[(size_t)iv * e.nu + iu] / e.marg_func_int / (2 * kPi * kPi * sinTheta);
}

struct LightSample {
    V3 wi;
    float pdf;
    bool black;      // Li == 0 (back-facing area light)
    float liScale;   // Li[bin] = L[bin] * liScale (point light: I / d^2 is a true division, flag below)
    bool divide;     // Li[bin] = (L[bin] * liMul) / liScale  (point: liMul = 1; spot: the falloff)
    float liMul;
    bool isEnv;      // infinite light: Li[bin] = IllumBin(env, bin)
    IllumRGB env;
    Interaction pLight;
};
template <unsigned TM>
DEV LightSample SampleLi(const DScene &s, const mi_light &l, const Interaction &ref, float u0, float u1) {
    LightSample ls;
    ls.pdf = 0; ls.black = true; ls.liScale = 1; ls.liMul = 1; ls.divide = false; ls.isEnv = false;
    ls.env.i1 = ls.env.i2 = 0; ls.env.w0 = ls.env.w1 = ls.env.w2 = 0;
    if (l.type == MI_LIGHT_DIFFUSE_AREA) {
        Interaction pShape = ShapeSample(s, l.shape, ref, u0, u1, &ls.pdf);
        if (ls.pdf == 0 || (pShape.p - ref.p).LengthSquared() == 0) { ls.pdf = 0; return ls; }
        ls.wi = Normalize(pShape.p - ref.p);
        ls.pLight = pShape;
        ls.black = !(l.two_sided || Dot(pShape.n, -ls.wi) > 0);
    } else if (TM_LIGHT(TM, MI_LIGHT_SPOT) && l.type == MI_LIGHT_SPOT) {  // spot.cpp:51-70
        V3 pLight(l.pos[0], l.pos[1], l.pos[2]);
        ls.wi = Normalize(pLight - ref.p);
        ls.pdf = 1.f;
        ls.pLight.p = pLight;
        V3 wl = Normalize(Mul3(l.w2l, -ls.wi));
        const float cosTheta = wl.z;
        float falloff;
        if (cosTheta < l.cos_total_width) falloff = 0;
        else if (cosTheta >= l.cos_falloff_start) falloff = 1;
        else {
            const float delta = (cosTheta - l.cos_total_width) / (l.cos_falloff_start - l.cos_total_width);
            falloff = (delta * delta) * (delta * delta);
        }
        ls.divide = true;
        ls.liMul = falloff;
        ls.liScale = DistanceSquared(pLight, ref.p);
        ls.black = false;   // (a zero falloff shows up bin by bin: the caller tests Li != 0)
    } else if (TM_LIGHT(TM, MI_LIGHT_INFINITE) && l.type == MI_LIGHT_INFINITE) {  // infinite.cpp:97-125
        const mi_envmap &e = s.envmaps[l.envmap];
        float pdfs[2];
        int v;
        const float d1 = SampleContinuous1D(e.marg_func, e.marg_cdf, e.marg_func_int, e.nv, u1, &pdfs[1], &v);
        const float d0 = SampleContinuous1D(e.cond_func + (size_t)v * e.nu, e.cond_cdf + (size_t)v * (e.nu + 1), e.cond_func_int[v],
                                            e.nu, u0, &pdfs[0], nullptr);
        const float mapPdf = pdfs[0] * pdfs[1];
        if (mapPdf == 0) return ls;
        const float theta = d1 * kPi, phi = d0 * 2 * kPi;
        const float cosTheta = cosF(theta), sinTheta = sinF(theta);
        const float sinPhi = sinF(phi), cosPhi = cosF(phi);
        ls.wi = Mul3(l.l2w, V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
        ls.pdf = mapPdf / (2 * kPi * kPi * sinTheta);
        if (sinTheta == 0) ls.pdf = 0;
        ls.pLight.p = ref.p + ls.wi * (2 * l.world_radius);
        float rgb[3];
        EnvLookup(e, d0, d1, rgb);
        ls.env = MakeIllumRGB(rgb);
        ls.isEnv = true;
        ls.black = false;   // (a black map value shows up bin by bin: the caller tests Li != 0)
    } else if (l.type == MI_LIGHT_POINT) {
        V3 pLight(l.pos[0], l.pos[1], l.pos[2]);
        ls.wi = Normalize(pLight - ref.p);
        ls.pdf = 1.f;
        ls.pLight.p = pLight;
        ls.black = false;
        ls.divide = true;
        ls.liScale = DistanceSquared(pLight, ref.p);
    } else {
        V3 wLight(l.dir[0], l.dir[1], l.dir[2]);
        ls.wi = wLight;
        ls.pdf = 1;
        ls.pLight.p = ref.p + wLight * (2 * l.world_radius);
        ls.black = false;
    }
    return ls;
}
template <unsigned TM>
DEV float LiBin(const DScene &s, const mi_light &l, const LightSample &ls, int bin) {
    if (TM_LIGHT(TM, MI_LIGHT_INFINITE) && ls.isEnv) return IllumBin(s, ls.env, bin);
    return ls.divide ? (l.L[bin] * ls.liMul) / ls.liScale : l.L[bin];
}

// Quad c (bins 4c..4c+3) of Li; the light's spectrum is one 16-B load (see EvalQuad).
template <unsigned TM>
DEV float4 LiQuad(const DScene &s, const mi_light &l, const LightSample &ls, int c) {
    if (TM_LIGHT(TM, MI_LIGHT_INFINITE) && ls.isEnv) {
        const int b = 4 * c;
        return make_float4(IllumBin(s, ls.env, b), IllumBin(s, ls.env, b + 1), IllumBin(s, ls.env, b + 2),
                           IllumBin(s, ls.env, min(b + 3, MI_NSPEC - 1)));
    }
    const float4 L = LoadSpec4(l.L, c);
    if (!ls.divide) return L;
    return make_float4((L.x * ls.liMul) / ls.liScale, (L.y * ls.liMul) / ls.liScale, (L.z * ls.liMul) / ls.liScale,
                       (L.w * ls.liMul) / ls.liScale);
}

// ------------------------------------------------------------------ Distribution1D
DEV int SampleDiscrete(const float *func, const float *cdf, float funcInt, int n, float u, float *pdf) {
    int size = n + 1;
    int first = 0, len = size;  // FindInterval, pbrt.h:405-418
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = first - 1;
    offset = offset < 0 ? 0 : (offset > size - 2 ? size - 2 : offset);
    *pdf = (funcInt > 0) ? func[offset] / (funcInt * n) : 0;
    return offset;
}
// Which distribution serves point p (lightdistrib.cpp:135-151; the hash table there is a cache).
DEV uint32_t LightDistribIndex(const DScene &s, const V3 &p) {
    if (s.ldType != MI_LD_SPATIAL) return 0;
    float o[3] = {p.x - s.wbMin[0], p.y - s.wbMin[1], p.z - s.wbMin[2]};
    int pi[3];
    for (int i = 0; i < 3; ++i) {
        if (s.wbMax[i] > s.wbMin[i]) o[i] /= s.wbMax[i] - s.wbMin[i];
        int v = (int)(o[i] * s.nVoxels[i]);
        pi[i] = v < 0 ? 0 : (v > s.nVoxels[i] - 1 ? s.nVoxels[i] - 1 : v);
    }
    return ((uint32_t)pi[2] * s.nVoxels[1] + pi[1]) * s.nVoxels[0] + pi[0];
}

}  // namespace dpt
