// oracle/oracle_pt.cpp -- TEST INFRASTRUCTURE: CPU oracle for the PathIntegrator
// hot path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this; the product (pbrt-v3-spectral_amd/) never links, imports or calls it.
//
// A restatement, against the flat mi_scene_desc, of
//   SamplerIntegrator::Render          src/core/integrator.cpp:228-342
//   PathIntegrator::Li                 src/integrators/path.cpp:64-188
//   UniformSampleOneLight/EstimateDirect  src/core/integrator.cpp:85-215
//   BVHAccel::Intersect/IntersectP     src/accelerators/bvh.cpp:662-738
//   GlobalSampler / HaltonSampler      src/core/sampler.cpp:46-52,136-195, src/samplers/halton.cpp:98-127
//   RadicalInverse / Scrambled...      src/core/lowdiscrepancy.cpp:389-424 and 130-175
//   PerspectiveCamera::GenerateRayDifferential  src/cameras/perspective.cpp:95-146
//   DiffuseAreaLight / Point / Distant src/lights/{diffuse,point,distant}.cpp
//   LightDistribution (uniform/power/spatial)   src/core/lightdistrib.cpp:48-300
//   FilmTile::AddSample / MergeFilmTile src/core/film.h:123-163, film.cpp:124-142
// Same depth-first, one-sample-at-a-time control flow as the reference (one tile per
// worker, samples of a pixel in order), so film sums accumulate in the same order.
//
// PARITY PINS: the reference cannot be built in this image without
// stand-ins for the absent glog submodule (src/core/pbrt.h:61), so no oracle/_ref
// exists; this oracle is pinned by the reference's own known-answer tests and by
// the deterministic counters the reference printed for killeroo-simple
// (BASELINE.md section 2), reproduced by tests/test_oracle_pins.py.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "../include/mi_pt.h"
#include "o_bsdf.h"
#include "o_math.h"
#include "o_shapes.h"

namespace orc {

struct Counters {
    uint64_t cameraRays = 0, regularRays = 0, shadowRays = 0, totalPaths = 0, zeroRadiancePaths = 0,
             pathLengthSum = 0, nodesVisited = 0, triTests = 0, badSamples = 0;
    void Add(const Counters &o) {
        cameraRays += o.cameraRays; regularRays += o.regularRays; shadowRays += o.shadowRays;
        totalPaths += o.totalPaths; zeroRadiancePaths += o.zeroRadiancePaths; pathLengthSum += o.pathLengthSum;
        nodesVisited += o.nodesVisited; triTests += o.triTests; badSamples += o.badSamples;
    }
};

// ------------------------------------------------------------------ sampler
inline uint32_t ReverseBits32(uint32_t n) {  // lowdiscrepancy.h:67-74
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
    return n;
}
inline uint64_t ReverseBits64(uint64_t n) {
    uint64_t n0 = ReverseBits32((uint32_t)n);
    uint64_t n1 = ReverseBits32((uint32_t)(n >> 32));
    return (n0 << 32) | n1;
}
static Float RadicalInverseBase(int base, uint64_t a) {  // RadicalInverseSpecialized<base>
    const Float invBase = (Float)1 / (Float)base;
    uint64_t reversedDigits = 0;
    Float invBaseN = 1;
    while (a) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversedDigits = reversedDigits * base + digit;
        invBaseN *= invBase;
        a = next;
    }
    return std::min(reversedDigits * invBaseN, OneMinusEpsilon);
}
static Float ScrambledRadicalInverseBase(int base, const uint16_t *perm, uint64_t a) {
    const Float invBase = (Float)1 / (Float)base;
    uint64_t reversedDigits = 0;
    Float invBaseN = 1;
    while (a) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversedDigits = reversedDigits * base + perm[digit];
        invBaseN *= invBase;
        a = next;
    }
    return std::min(invBaseN * (reversedDigits + invBase * perm[0] / (1 - invBase)), OneMinusEpsilon);
}
static Float RadicalInverse(const mi_scene_desc &d, int baseIndex, uint64_t a) {  // lowdiscrepancy.cpp:389-424
    if (baseIndex == 0) return ReverseBits64(a) * 0x1p-64;
    return RadicalInverseBase(d.sampler.primes[baseIndex], a);
}
template <int base>
inline uint64_t InverseRadicalInverse(uint64_t inverse, int nDigits) {  // lowdiscrepancy.h:82-91
    uint64_t index = 0;
    for (int i = 0; i < nDigits; ++i) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}
template <typename T>
inline T Mod(T a, T b) { T result = a - (a / b) * b; return (T)((result < 0) ? result + b : result); }

// SobolIntervalToIndex, lowdiscrepancy.h:229-249 (the two pixel-index matrices of the film's resolution travel in mi_sampler)
static uint64_t SobolIntervalToIndex(const mi_sampler &s, uint64_t frame, int px, int py) {
    const uint32_t m = (uint32_t)s.sobol_log2_resolution;
    if (m == 0) return 0;
    const uint32_t m2 = m << 1;
    uint64_t index = uint64_t(frame) << m2;
    uint64_t delta = 0;
    for (int c = 0; frame; frame >>= 1, ++c)
        if (frame & 1) delta ^= s.sobol_vdc[c];
    uint64_t b = (((uint64_t)((uint32_t)px) << m) | ((uint32_t)py)) ^ delta;
    for (int c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= s.sobol_vdc_inv[c];
    return index;
}
// SobolSampleFloat, lowdiscrepancy.h:259-274 (no scrambling: SobolSampler::SampleDimension passes none)
static float SobolSampleFloat(const mi_sampler &s, int64_t a, int dimension) {
    uint32_t v = 0;
    for (int i = dimension * MI_SOBOL_MATRIX_SIZE; a != 0; a >>= 1, i++)
        if (a & 1) v ^= s.sobol_matrices[i];
    return std::min(v * 0x1p-32f, OneMinusEpsilon);
}

struct PCG32 {  // RNG, rng.h:61-144
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    void SetSequence(uint64_t initseq) {
        state = 0u;
        inc = (initseq << 1u) | 1u;
        UniformUInt32();
        state += 0x853c49e6748fea9bULL;
        UniformUInt32();
    }
    uint32_t UniformUInt32() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
        uint32_t rot = (uint32_t)(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    uint32_t UniformUInt32(uint32_t b) {   // rng.h:112-121
        uint32_t threshold = (~b + 1u) % b;
        while (true) {
            uint32_t r = UniformUInt32();
            if (r >= threshold) return r % b;
        }
    }
    Float UniformFloat() { return std::min(OneMinusEpsilon, Float(UniformUInt32() * 0x1p-32f)); }
};

// ---- the pixel samplers' tables (mi_sampler_type: ZEROTWO, STRATIFIED)
// Shuffle, sampling.h:150-157
template <typename T>
static void Shuffle(T *samp, int count, int nDimensions, PCG32 &rng) {
    for (int i = 0; i < count; ++i) {
        int other = i + rng.UniformUInt32(count - i);
        for (int j = 0; j < nDimensions; ++j) std::swap(samp[nDimensions * i + j], samp[nDimensions * other + j]);
    }
}
// MultiplyGenerator / SampleGeneratorMatrix / GrayCodeSample, lowdiscrepancy.h:93-147
static uint32_t MultiplyGenerator(const uint32_t *C, uint32_t a) {
    uint32_t v = 0;
    for (int i = 0; a != 0; ++i, a >>= 1)
        if (a & 1) v ^= C[i];
    return v;
}
static Float SampleGeneratorMatrix(const uint32_t *C, uint32_t a, uint32_t scramble = 0) {
    return std::min((MultiplyGenerator(C, a) ^ scramble) * Float(0x1p-32), OneMinusEpsilon);
}
static void GrayCodeSample(const uint32_t *C, uint32_t n, uint32_t scramble, Float *p) {
    uint32_t v = scramble;
    for (uint32_t i = 0; i < n; ++i) {
        p[i] = std::min(v * Float(0x1p-32), OneMinusEpsilon);
        v ^= C[__builtin_ctz(i + 1)];
    }
}
static void GrayCodeSample2(const uint32_t *C0, const uint32_t *C1, uint32_t n, const uint32_t scramble[2], Float *p /* x, y pairs */) {
    uint32_t v[2] = {scramble[0], scramble[1]};
    for (uint32_t i = 0; i < n; ++i) {
        p[2 * i] = std::min(v[0] * Float(0x1p-32), OneMinusEpsilon);
        p[2 * i + 1] = std::min(v[1] * Float(0x1p-32), OneMinusEpsilon);
        v[0] ^= C0[__builtin_ctz(i + 1)];
        v[1] ^= C1[__builtin_ctz(i + 1)];
    }
}
// The generator matrices: van der Corput's is the bit-reversed identity (lowdiscrepancy.h:155-203), the second Sobol' one
// Pascal's triangle mod 2 (lowdiscrepancy.h:208-224) -- column i of it is column i - 1 XOR itself shifted right by one.
struct ZeroTwoMatrices {
    uint32_t vdc[32], sobol1[32];
    ZeroTwoMatrices() {
        for (int i = 0; i < 32; ++i) vdc[i] = 0x80000000u >> i;
        sobol1[0] = 0x80000000u;
        for (int i = 1; i < 32; ++i) sobol1[i] = sobol1[i - 1] ^ (sobol1[i - 1] >> 1);
    }
};
static const ZeroTwoMatrices kZeroTwo;
// VanDerCorput / Sobol2D, lowdiscrepancy.h:149-227
static void VanDerCorput(int nSamplesPerPixelSample, int nPixelSamples, Float *samples, PCG32 &rng) {
    uint32_t scramble = rng.UniformUInt32();
    int totalSamples = nSamplesPerPixelSample * nPixelSamples;
    GrayCodeSample(kZeroTwo.vdc, totalSamples, scramble, samples);
    for (int i = 0; i < nPixelSamples; ++i) Shuffle(samples + i * nSamplesPerPixelSample, nSamplesPerPixelSample, 1, rng);
    Shuffle(samples, nPixelSamples, nSamplesPerPixelSample, rng);
}
static void Sobol2D(int nSamplesPerPixelSample, int nPixelSamples, Float *samples /* pairs */, PCG32 &rng) {
    uint32_t scramble[2];
    scramble[0] = rng.UniformUInt32();
    scramble[1] = rng.UniformUInt32();
    GrayCodeSample2(kZeroTwo.vdc, kZeroTwo.sobol1, nSamplesPerPixelSample * nPixelSamples, scramble, samples);
    struct P2 { Float x, y; };
    P2 *sp = reinterpret_cast<P2 *>(samples);
    for (int i = 0; i < nPixelSamples; ++i) Shuffle(sp + i * nSamplesPerPixelSample, nSamplesPerPixelSample, 1, rng);
    Shuffle(sp, nPixelSamples, nSamplesPerPixelSample, rng);
}
// StratifiedSample1D / 2D, sampling.cpp:42-60
static void StratifiedSample1D(Float *samp, int nSamples, PCG32 &rng, bool jitter) {
    Float invNSamples = (Float)1 / nSamples;
    for (int i = 0; i < nSamples; ++i) {
        Float delta = jitter ? rng.UniformFloat() : 0.5f;
        samp[i] = std::min((i + delta) * invNSamples, OneMinusEpsilon);
    }
}
static void StratifiedSample2D(Float *samp /* pairs */, int nx, int ny, PCG32 &rng, bool jitter) {
    Float dx = (Float)1 / nx, dy = (Float)1 / ny;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            Float jx = jitter ? rng.UniformFloat() : 0.5f;
            Float jy = jitter ? rng.UniformFloat() : 0.5f;
            samp[0] = std::min((x + jx) * dx, OneMinusEpsilon);
            samp[1] = std::min((y + jy) * dy, OneMinusEpsilon);
            samp += 2;
        }
}

struct Sampler {  // GlobalSampler + HaltonSampler / SobolSampler state for one pixel sample; RandomSampler with one stream per sample
    const mi_scene_desc &d;
    int64_t offsetForCurrentPixel = 0;
    int64_t intervalSampleIndex = 0;
    int dimension = 0;
    int curPx = 0, curPy = 0;
    PCG32 rng;
    // PixelSampler (sampler.cpp:100-135): the pixel's tables, the sample in hand, the next 1D / 2D table
    std::vector<std::vector<Float>> samples1D, samples2D;   // [dimension][sample] / [dimension][2 * sample]
    int64_t curSample = 0;
    int current1DDimension = 0, current2DDimension = 0;
    bool IsPixelSampler() const { return d.sampler.type == MI_SAMPLER_ZEROTWO || d.sampler.type == MI_SAMPLER_STRATIFIED; }
    int64_t PixelIndex() const {
        const int64_t w = d.film.sample_bounds[2] - d.film.sample_bounds[0];
        return (int64_t)(curPy - d.film.sample_bounds[1]) * w + (curPx - d.film.sample_bounds[0]);
    }
    int64_t PixelCount() const {
        return (int64_t)(d.film.sample_bounds[2] - d.film.sample_bounds[0]) * (d.film.sample_bounds[3] - d.film.sample_bounds[1]);
    }
    explicit Sampler(const mi_scene_desc &d) : d(d) {}
    // ZeroTwoSequenceSampler::StartPixel (zerotwosequence.cpp:53-69) / StratifiedSampler::StartPixel (stratified.cpp:43-71), the
    // pixel's own stream standing in for the tile's (mi_sampler_type); no sample arrays are requested on this path
    void StartPixelTables() {
        const mi_sampler &s = d.sampler;
        const int spp = (int)s.samples_per_pixel;
        PCG32 prng;
        prng.SetSequence((uint64_t)PixelIndex());
        samples1D.assign(s.pixel_dims, std::vector<Float>(spp));
        samples2D.assign(s.pixel_dims, std::vector<Float>(2 * (size_t)spp));
        struct P2 { Float x, y; };
        if (s.type == MI_SAMPLER_ZEROTWO) {
            for (auto &v : samples1D) VanDerCorput(1, spp, v.data(), prng);
            for (auto &v : samples2D) Sobol2D(1, spp, v.data(), prng);
        } else {
            for (auto &v : samples1D) {
                StratifiedSample1D(v.data(), s.x_samples * s.y_samples, prng, s.jitter != 0);
                Shuffle(v.data(), s.x_samples * s.y_samples, 1, prng);
            }
            for (auto &v : samples2D) {
                StratifiedSample2D(v.data(), s.x_samples, s.y_samples, prng, s.jitter != 0);
                Shuffle(reinterpret_cast<P2 *>(v.data()), s.x_samples * s.y_samples, 1, prng);
            }
        }
    }
    void StartPixel(int px, int py) {  // halton.cpp:98-118 (offset part of GetIndexForSample)
        const mi_sampler &s = d.sampler;
        curPx = px; curPy = py;
        offsetForCurrentPixel = 0;
        if (IsPixelSampler()) { StartPixelTables(); return; }
        if (s.type != MI_SAMPLER_HALTON) return;
        const int kMaxResolution = 128;
        if (s.sample_stride > 1) {
            int pm[2] = {Mod(px, kMaxResolution), Mod(py, kMaxResolution)};
            for (int i = 0; i < 2; ++i) {
                uint64_t dimOffset = (i == 0) ? InverseRadicalInverse<2>(pm[i], s.base_exponents[i])
                                              : InverseRadicalInverse<3>(pm[i], s.base_exponents[i]);
                offsetForCurrentPixel += dimOffset * (s.sample_stride / s.base_scales[i]) * s.mult_inverse[i];
            }
            offsetForCurrentPixel %= s.sample_stride;
        }
    }
    void StartSample(int64_t sampleNum) {
        const mi_sampler &s = d.sampler;
        dimension = 0;
        if (IsPixelSampler()) {   // PixelSampler::SetSampleNumber; the sample's own stream for the draws beyond the tables
            curSample = sampleNum;
            current1DDimension = current2DDimension = 0;
            rng.SetSequence((uint64_t)((sampleNum + 1) * PixelCount() + PixelIndex()));
            return;
        }
        if (s.type == MI_SAMPLER_SOBOL)   // SobolSampler::GetIndexForSample, sobol.cpp:42-45
            intervalSampleIndex = (int64_t)SobolIntervalToIndex(s, (uint64_t)sampleNum, curPx - d.film.sample_bounds[0], curPy - d.film.sample_bounds[1]);
        else if (s.type == MI_SAMPLER_RANDOM) {   // one stream per camera sample (see mi_sampler_type)
            const int64_t w = d.film.sample_bounds[2] - d.film.sample_bounds[0], h = d.film.sample_bounds[3] - d.film.sample_bounds[1];
            const int64_t pix = (int64_t)(curPy - d.film.sample_bounds[1]) * w + (curPx - d.film.sample_bounds[0]);
            rng.SetSequence((uint64_t)(sampleNum * (w * h) + pix));   // (distinct for every pixel and sample number of any pass)
        } else intervalSampleIndex = offsetForCurrentPixel + sampleNum * s.sample_stride;
    }
    Float SampleDimension(int64_t index, int dim) const {  // halton.cpp:120-127 / sobol.cpp:47-59
        const mi_sampler &s = d.sampler;
        if (s.type == MI_SAMPLER_SOBOL) {
            Float v = SobolSampleFloat(s, index, dim);
            if (dim == 0 || dim == 1) {   // remap the dimensions used for the pixel sample
                v = v * s.sobol_resolution + d.film.sample_bounds[dim];
                v = Clamp(v - (dim == 0 ? curPx : curPy), (Float)0, OneMinusEpsilon);
            }
            return v;
        }
        if (s.sample_at_pixel_center && (dim == 0 || dim == 1)) return 0.5f;
        if (dim == 0) return RadicalInverse(d, dim, index >> s.base_exponents[0]);
        else if (dim == 1) return RadicalInverse(d, dim, index / s.base_scales[1]);
        else return ScrambledRadicalInverseBase(s.primes[dim], &s.perms[s.prime_sums[dim]], index);
    }
    // arrayStartDim == arrayEndDim == 5 (no sample arrays requested): no skipping, sampler.cpp:178-195
    Float Get1D() {
        if (IsPixelSampler()) {   // PixelSampler::Get1D, sampler.cpp:119-126
            if (current1DDimension < (int)samples1D.size()) return samples1D[current1DDimension++][curSample];
            return rng.UniformFloat();
        }
        if (d.sampler.type == MI_SAMPLER_RANDOM) { ++dimension; return rng.UniformFloat(); }   // random.cpp:44-48
        return SampleDimension(intervalSampleIndex, dimension++);
    }
    void Get2D(Float u[2]) {
        if (IsPixelSampler()) {   // PixelSampler::Get2D, sampler.cpp:128-135
            if (current2DDimension < (int)samples2D.size()) {
                const std::vector<Float> &v = samples2D[current2DDimension++];
                u[0] = v[2 * curSample]; u[1] = v[2 * curSample + 1];
            } else { u[0] = rng.UniformFloat(); u[1] = rng.UniformFloat(); }
            return;
        }
        if (d.sampler.type == MI_SAMPLER_RANDOM) { dimension += 2; u[0] = rng.UniformFloat(); u[1] = rng.UniformFloat(); return; }   // random.cpp:50-54
        u[0] = SampleDimension(intervalSampleIndex, dimension);
        u[1] = SampleDimension(intervalSampleIndex, dimension + 1);
        dimension += 2;
    }
};

// ------------------------------------------------------------------ scene intersection
struct Scene {
    const mi_scene_desc &d;
    explicit Scene(const mi_scene_desc &d) : d(d) {}

    static bool BoundsIntersectP(const mi_bvh_node &n, const Ray &ray, const V3 &invDir, const int dirIsNeg[3]) {
        // geometry.h:1420-1447
        auto b = [&](int i, int axis) { return i ? n.bmax[axis] : n.bmin[axis]; };
        Float tMin = (b(dirIsNeg[0], 0) - ray.o.x) * invDir.x;
        Float tMax = (b(1 - dirIsNeg[0], 0) - ray.o.x) * invDir.x;
        Float tyMin = (b(dirIsNeg[1], 1) - ray.o.y) * invDir.y;
        Float tyMax = (b(1 - dirIsNeg[1], 1) - ray.o.y) * invDir.y;
        tMax *= 1 + 2 * gamma(3);
        tyMax *= 1 + 2 * gamma(3);
        if (tMin > tyMax || tyMin > tMax) return false;
        if (tyMin > tMin) tMin = tyMin;
        if (tyMax < tMax) tMax = tyMax;
        Float tzMin = (b(dirIsNeg[2], 2) - ray.o.z) * invDir.z;
        Float tzMax = (b(1 - dirIsNeg[2], 2) - ray.o.z) * invDir.z;
        tzMax *= 1 + 2 * gamma(3);
        if (tMin > tzMax || tzMin > tMax) return false;
        if (tzMin > tMin) tMin = tzMin;
        if (tzMax < tMax) tMax = tzMax;
        return (tMin < ray.tMax) && (tMax > 0);
    }

    // Transform::operator()(const SurfaceInteraction&), transform.cpp:262-297, with InstanceToWorld (w2i = its stored inverse)
    static void ToWorld(const mi_instance &in, SurfaceInteraction *si) {
        const float *m = in.i2w, *mInv = in.w2i;
        V3 pErr;
        si->p = XfPointErr2(m, si->p, si->pError, &pErr);
        si->pError = pErr;
        si->n = Normalize(XfNormal(mInv, si->n));
        si->wo = Normalize(XfVector(m, si->wo));
        si->dpdu = XfVector(m, si->dpdu);
        si->dpdv = XfVector(m, si->dpdv);
        si->dndu = XfNormal(mInv, si->dndu);
        si->dndv = XfNormal(mInv, si->dndv);
        si->shading.n = Normalize(XfNormal(mInv, si->shading.n));
        si->shading.dpdu = XfVector(m, si->shading.dpdu);
        si->shading.dpdv = XfVector(m, si->shading.dpdv);
        si->shading.dndu = XfNormal(mInv, si->shading.dndu);
        si->shading.dndv = XfNormal(mInv, si->shading.dndv);
        si->shading.n = Faceforward(si->shading.n, si->n);
    }

    // GeometricPrimitive::Intersect, primitive.cpp:119-135; TransformedPrimitive::Intersect, primitive.cpp:78-92
    bool PrimIntersect(int primIdx, const Ray &ray, SurfaceInteraction *isect, Counters &c) const {
        const mi_prim &p = d.prims[primIdx];
        if (p.instance > 0) {
            const mi_instance &in = d.instances[p.instance - 1];
            Ray r2 = XfRay(in.w2i, ray);   // Inverse(InstanceToWorld)(r)
            if (!BVHIntersect((int)in.root, r2, isect, c)) return false;
            ray.tMax = r2.tMax;
            ToWorld(in, isect);
            return true;
        }
        Float tHit;
        if (p.shape >= 0) {
            ++c.triTests;
            if (!TriIntersect(d, p.shape, ray, &tHit, isect)) return false;
        } else {
            if (!SphereIntersect(d.spheres[~p.shape], ray, &tHit, isect)) return false;
        }
        ray.tMax = tHit;
        isect->prim = primIdx;
        return true;
    }
    bool PrimIntersectP(int primIdx, const Ray &ray, Counters &c) const {
        const mi_prim &p = d.prims[primIdx];
        if (p.instance > 0) {   // TransformedPrimitive::IntersectP, primitive.cpp:94-99
            const mi_instance &in = d.instances[p.instance - 1];
            return BVHIntersectP((int)in.root, XfRay(in.w2i, ray), c);
        }
        if (p.shape >= 0) {
            ++c.triTests;
            return TriIntersectP(d, p.shape, ray);
        }
        return SphereIntersectP(d.spheres[~p.shape], ray);
    }

    bool Intersect(const Ray &ray, SurfaceInteraction *isect, Counters &c) const {  // scene.cpp:45-49
        ++c.regularRays;
        if (d.n_nodes == 0) return false;
        return BVHIntersect(0, ray, isect, c);
    }
    bool IntersectP(const Ray &ray, Counters &c) const {  // scene.cpp:51-55
        ++c.shadowRays;
        if (d.n_nodes == 0) return false;
        return BVHIntersectP(0, ray, c);
    }

    bool BVHIntersect(int root, const Ray &ray, SurfaceInteraction *isect, Counters &c) const {  // bvh.cpp:662-700
        bool hit = false;
        V3 invDir(1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z);
        int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
        int toVisitOffset = 0, currentNodeIndex = root;
        int nodesToVisit[64];
        while (true) {
            const mi_bvh_node *node = &d.nodes[currentNodeIndex];
            ++c.nodesVisited;
            if (BoundsIntersectP(*node, ray, invDir, dirIsNeg)) {
                if (node->n_prims > 0) {
                    for (int i = 0; i < node->n_prims; ++i)
                        if (PrimIntersect(node->offset + i, ray, isect, c)) hit = true;
                    if (toVisitOffset == 0) break;
                    currentNodeIndex = nodesToVisit[--toVisitOffset];
                } else {
                    if (dirIsNeg[node->axis]) {
                        nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
                        currentNodeIndex = node->offset;
                    } else {
                        nodesToVisit[toVisitOffset++] = node->offset;
                        currentNodeIndex = currentNodeIndex + 1;
                    }
                }
            } else {
                if (toVisitOffset == 0) break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            }
        }
        return hit;
    }
    bool BVHIntersectP(int root, const Ray &ray, Counters &c) const {  // bvh.cpp:702-738
        V3 invDir(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
        int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
        int nodesToVisit[64];
        int toVisitOffset = 0, currentNodeIndex = root;
        while (true) {
            const mi_bvh_node *node = &d.nodes[currentNodeIndex];
            ++c.nodesVisited;
            if (BoundsIntersectP(*node, ray, invDir, dirIsNeg)) {
                if (node->n_prims > 0) {
                    for (int i = 0; i < node->n_prims; ++i)
                        if (PrimIntersectP(node->offset + i, ray, c)) return true;
                    if (toVisitOffset == 0) break;
                    currentNodeIndex = nodesToVisit[--toVisitOffset];
                } else {
                    if (dirIsNeg[node->axis]) {
                        nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
                        currentNodeIndex = node->offset;
                    } else {
                        nodesToVisit[toVisitOffset++] = node->offset;
                        currentNodeIndex = currentNodeIndex + 1;
                    }
                }
            } else {
                if (toVisitOffset == 0) break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            }
        }
        return false;
    }
};

// ------------------------------------------------------------------ lights

struct LightSample {
    Spec Li;
    V3 wi;
    Float pdf = 0;
    Interaction pLight;  // VisibilityTester p1
};

// Shape::Sample(ref,u,pdf) for the light's shape: sphere.cpp:232-292 or shape.cpp:56-70.
static Interaction ShapeSample(const mi_scene_desc &d, int shape, const Interaction &ref, const Float u[2], Float *pdf) {
    if (shape < 0) return SphereSample(d.spheres[~shape], ref, u, pdf);
    Interaction intr = TriSample(d, shape, u, pdf);
    V3 wi = intr.p - ref.p;
    if (wi.LengthSquared() == 0) *pdf = 0;
    else {
        wi = Normalize(wi);
        *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
        if (std::isinf(*pdf)) *pdf = 0.f;
    }
    return intr;
}
// Shape::Pdf(ref, wi): sphere.cpp:294-306 / shape.cpp:72-87
static Float ShapePdf(const mi_scene_desc &d, int shape, Float area, const Interaction &ref, const V3 &wi) {
    if (shape < 0) {
        const mi_sphere &s = d.spheres[~shape];
        V3 pCenter = XfPoint(s.o2w, V3(0, 0, 0));
        V3 pOrigin = OffsetRayOrigin(ref.p, ref.pError, ref.n, pCenter - ref.p);
        if (!(DistanceSquared(pOrigin, pCenter) <= s.radius * s.radius)) {
            Float sinThetaMax2 = s.radius * s.radius / DistanceSquared(ref.p, pCenter);
            Float cosThetaMax = std::sqrt(std::max((Float)0, 1 - sinThetaMax2));
            return 1 / (2 * Pi * (1 - cosThetaMax));  // UniformConePdf
        }
    }
    Ray ray = SpawnRay(ref, wi);
    Float tHit;
    SurfaceInteraction isectLight;
    bool hit = (shape < 0) ? SphereIntersect(d.spheres[~shape], ray, &tHit, &isectLight)
                           : TriIntersect(d, shape, ray, &tHit, &isectLight);
    if (!hit) return 0;
    Float pdf = DistanceSquared(ref.p, isectLight.p) / (AbsDot(isectLight.n, -wi) * area);
    if (std::isinf(pdf)) pdf = 0.f;
    return pdf;
}

static bool IsDeltaLight(const mi_light &l) { return l.type == MI_LIGHT_POINT || l.type == MI_LIGHT_DISTANT || l.type == MI_LIGHT_SPOT; }

// ---- InfiniteAreaLight, src/lights/infinite.cpp:85-144
// Lmap->Lookup(st) with width 0: level < 0 -> triangle(0, st), bilinear with ImageWrap::Repeat (mipmap.h:252-281)
static void EnvLookup(const mi_envmap &e, const Float st[2], Float rgb[3]) {
    Float s = st[0] * e.width - 0.5f;
    Float t = st[1] * e.height - 0.5f;
    int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
    Float ds = s - s0, dt = t - t0;
    auto texel = [&](int ss, int tt, int k) { return e.rgb[3 * ((size_t)Mod(tt, e.height) * e.width + Mod(ss, e.width)) + k]; };
    for (int k = 0; k < 3; ++k)
        rgb[k] = (1 - ds) * (1 - dt) * texel(s0, t0, k) + (1 - ds) * dt * texel(s0, t0 + 1, k) + ds * (1 - dt) * texel(s0 + 1, t0, k) +
                 ds * dt * texel(s0 + 1, t0 + 1, k);
}
// Spectrum(rgb, SpectrumType::Illuminant): SampledSpectrum::FromRGB, spectrum.cpp:98-180
static V3 Mul3(const float m[9], const V3 &v) {  // Transform::operator()(Vector3f), transform.h:235-240
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
static Float SphericalTheta(const V3 &v) { return AcosF(Clamp(v.z, -1, 1)); }
static Float SphericalPhi(const V3 &v) { Float p = Atan2F(v.y, v.x); return (p < 0) ? (p + 2 * Pi) : p; }
static Spec InfiniteLe(const mi_scene_desc &d, const mi_light &l, const V3 &dir) {  // infinite.cpp:91-95
    V3 w = Normalize(Mul3(l.w2l, dir));
    Float st[2] = {SphericalPhi(w) * Inv2Pi, SphericalTheta(w) * InvPi}, rgb[3];
    EnvLookup(d.envmaps[l.envmap], st, rgb);
    return SpecFromRGBIllum(d, rgb);
}
// Distribution1D::SampleContinuous over tabulated func/cdf, sampling.h:71-89
static Float SampleContinuous1D(const float *func, const float *cdf, Float funcInt, int n, Float u, Float *pdf, int *off) {
    int size = n + 1, first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = Clamp(first - 1, 0, size - 2);
    if (off) *off = offset;
    Float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    if (pdf) *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / n;
}
static Float InfinitePdfLi(const mi_scene_desc &d, const mi_light &l, const V3 &w) {  // infinite.cpp:133-141
    const mi_envmap &e = d.envmaps[l.envmap];
    V3 wi = Mul3(l.w2l, w);
    Float theta = SphericalTheta(wi), phi = SphericalPhi(wi);
    Float sinTheta = SinF(theta);
    if (sinTheta == 0) return 0;
    Float p[2] = {phi * Inv2Pi, theta * InvPi};
    int iu = Clamp(int(p[0] * e.nu), 0, e.nu - 1);   // Distribution2D::Pdf, sampling.h:137-143
    int iv = Clamp(int(p[1] * e.nv), 0, e.nv - 1);
    return e.cond_func[(size_t)iv * e.nu + iu] / e.marg_func_int / (2 * Pi * Pi * sinTheta);
}

static LightSample SampleLi(const mi_scene_desc &d, const mi_light &l, const Interaction &ref, const Float u[2]) {
    LightSample ls;
    if (l.type == MI_LIGHT_DIFFUSE_AREA) {  // diffuse.cpp:68-81
        Interaction pShape = ShapeSample(d, l.shape, ref, u, &ls.pdf);
        if (ls.pdf == 0 || (pShape.p - ref.p).LengthSquared() == 0) {
            ls.pdf = 0;
            ls.Li = Spec(0.f);
            return ls;
        }
        ls.wi = Normalize(pShape.p - ref.p);
        ls.pLight = pShape;
        ls.Li = (l.two_sided || Dot(pShape.n, -ls.wi) > 0) ? Spec::From(l.L) : Spec(0.f);  // diffuse.h:56-58
    } else if (l.type == MI_LIGHT_POINT) {  // point.cpp:44-53
        V3 pLight(l.pos[0], l.pos[1], l.pos[2]);
        ls.wi = Normalize(pLight - ref.p);
        ls.pdf = 1.f;
        ls.pLight = Interaction();
        ls.pLight.p = pLight;
        ls.Li = Spec::From(l.L) / DistanceSquared(pLight, ref.p);
    } else if (l.type == MI_LIGHT_SPOT) {  // spot.cpp:51-70
        V3 pLight(l.pos[0], l.pos[1], l.pos[2]);
        ls.wi = Normalize(pLight - ref.p);
        ls.pdf = 1.f;
        ls.pLight = Interaction();
        ls.pLight.p = pLight;
        V3 wl = Normalize(Mul3(l.w2l, -ls.wi));
        Float cosTheta = wl.z, falloff;
        if (cosTheta < l.cos_total_width) falloff = 0;
        else if (cosTheta >= l.cos_falloff_start) falloff = 1;
        else {
            Float delta = (cosTheta - l.cos_total_width) / (l.cos_falloff_start - l.cos_total_width);
            falloff = (delta * delta) * (delta * delta);
        }
        ls.Li = Spec::From(l.L) * falloff / DistanceSquared(pLight, ref.p);
    } else if (l.type == MI_LIGHT_INFINITE) {  // infinite.cpp:97-125
        const mi_envmap &e = d.envmaps[l.envmap];
        ls.pdf = 0;
        ls.Li = Spec(0.f);
        ls.pLight = Interaction();
        Float pdfs[2];
        int v;
        Float d1 = SampleContinuous1D(e.marg_func, e.marg_cdf, e.marg_func_int, e.nv, u[1], &pdfs[1], &v);
        Float d0 = SampleContinuous1D(e.cond_func + (size_t)v * e.nu, e.cond_cdf + (size_t)v * (e.nu + 1), e.cond_func_int[v], e.nu, u[0],
                                      &pdfs[0], nullptr);
        Float mapPdf = pdfs[0] * pdfs[1];
        if (mapPdf == 0) return ls;
        Float theta = d1 * Pi, phi = d0 * 2 * Pi;
        Float cosTheta = CosF(theta), sinTheta = SinF(theta);
        Float sinPhi = SinF(phi), cosPhi = CosF(phi);
        ls.wi = Mul3(l.l2w, V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
        ls.pdf = mapPdf / (2 * Pi * Pi * sinTheta);
        if (sinTheta == 0) ls.pdf = 0;
        ls.pLight.p = ref.p + ls.wi * (2 * l.world_radius);
        Float uv[2] = {d0, d1}, rgb[3];
        EnvLookup(e, uv, rgb);
        ls.Li = SpecFromRGBIllum(d, rgb);
    } else {  // distant.cpp:49-59
        V3 wLight(l.dir[0], l.dir[1], l.dir[2]);
        ls.wi = wLight;
        ls.pdf = 1;
        ls.pLight = Interaction();
        ls.pLight.p = ref.p + wLight * (2 * l.world_radius);
        ls.Li = Spec::From(l.L);
    }
    return ls;
}

// ------------------------------------------------------------------ light distribution
struct Distribution1D {  // sampling.h:55-109
    std::vector<Float> func, cdf;
    Float funcInt;
    Distribution1D(const Float *f, int n) : func(f, f + n), cdf(n + 1) {
        cdf[0] = 0;
        for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
        funcInt = cdf[n];
        if (funcInt == 0) { for (int i = 1; i < n + 1; ++i) cdf[i] = Float(i) / Float(n); }
        else { for (int i = 1; i < n + 1; ++i) cdf[i] /= funcInt; }
    }
    int Count() const { return (int)func.size(); }
    int SampleDiscrete(Float u, Float *pdf) const {
        int size = (int)cdf.size();
        int first = 0, len = size;  // FindInterval, pbrt.h:405-418
        while (len > 0) {
            int half = len >> 1, middle = first + half;
            if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        int offset = Clamp(first - 1, 0, size - 2);
        if (pdf) *pdf = (funcInt > 0) ? func[offset] / (funcInt * Count()) : 0;
        return offset;
    }
};

struct LightDistribution {
    const mi_scene_desc &d;
    std::unique_ptr<Distribution1D> single;
    std::vector<std::atomic<Distribution1D *>> voxels;
    std::mutex mu;
    V3 bmin, bmax;
    explicit LightDistribution(const mi_scene_desc &d) : d(d) {
        const mi_lightdistrib &ld = d.light_distrib;
        if (d.n_lights == 0) return;
        if (ld.type != MI_LD_SPATIAL) {
            single.reset(new Distribution1D(ld.func, (int)d.n_lights));
        } else {
            size_t n = (size_t)ld.n_voxels[0] * ld.n_voxels[1] * ld.n_voxels[2];
            voxels = std::vector<std::atomic<Distribution1D *>>(n);
            for (auto &v : voxels) v.store(nullptr);
            bmin = V3(d.nodes[0].bmin[0], d.nodes[0].bmin[1], d.nodes[0].bmin[2]);
            bmax = V3(d.nodes[0].bmax[0], d.nodes[0].bmax[1], d.nodes[0].bmax[2]);
        }
    }
    ~LightDistribution() { for (auto &v : voxels) delete v.load(); }

    V3 BoundsLerp(const V3 &t) const {
        return V3(Lerp(t.x, bmin.x, bmax.x), Lerp(t.y, bmin.y, bmax.y), Lerp(t.z, bmin.z, bmax.z));
    }
    Distribution1D *ComputeDistribution(const int pi[3]) const {  // lightdistrib.cpp:232-300
        const int *nVoxels = d.light_distrib.n_voxels;
        V3 p0(Float(pi[0]) / Float(nVoxels[0]), Float(pi[1]) / Float(nVoxels[1]), Float(pi[2]) / Float(nVoxels[2]));
        V3 p1(Float(pi[0] + 1) / Float(nVoxels[0]), Float(pi[1] + 1) / Float(nVoxels[1]),
              Float(pi[2] + 1) / Float(nVoxels[2]));
        V3 vb0 = BoundsLerp(p0), vb1 = BoundsLerp(p1);
        V3 vmin(std::min(vb0.x, vb1.x), std::min(vb0.y, vb1.y), std::min(vb0.z, vb1.z));
        V3 vmax(std::max(vb0.x, vb1.x), std::max(vb0.y, vb1.y), std::max(vb0.z, vb1.z));
        int nSamples = 128;
        std::vector<Float> lightContrib(d.n_lights, Float(0));
        for (int i = 0; i < nSamples; ++i) {
            V3 t(RadicalInverse(d, 0, i), RadicalInverse(d, 1, i), RadicalInverse(d, 2, i));
            V3 po(Lerp(t.x, vmin.x, vmax.x), Lerp(t.y, vmin.y, vmax.y), Lerp(t.z, vmin.z, vmax.z));
            Interaction intr;
            intr.p = po;
            intr.wo = V3(1, 0, 0);
            Float u[2] = {RadicalInverse(d, 3, i), RadicalInverse(d, 4, i)};
            for (uint32_t j = 0; j < d.n_lights; ++j) {
                LightSample ls = SampleLi(d, d.lights[j], intr, u);
                if (ls.pdf > 0) lightContrib[j] += SpecY(d, ls.Li) / ls.pdf;
            }
        }
        Float sumContrib = 0;
        for (Float v : lightContrib) sumContrib += v;  // std::accumulate(..., Float(0))
        Float avgContrib = sumContrib / (nSamples * lightContrib.size());
        Float minContrib = (avgContrib > 0) ? .001 * avgContrib : 1;
        for (size_t i = 0; i < lightContrib.size(); ++i) lightContrib[i] = std::max(lightContrib[i], minContrib);
        return new Distribution1D(&lightContrib[0], int(lightContrib.size()));
    }
    const Distribution1D *Lookup(const V3 &p) {  // lightdistrib.cpp:135-230 (hash table == cache only)
        if (single) return single.get();
        if (d.n_lights == 0) return nullptr;
        const int *nVoxels = d.light_distrib.n_voxels;
        V3 o = p - bmin;  // Bounds3::Offset
        if (bmax.x > bmin.x) o.x /= bmax.x - bmin.x;
        if (bmax.y > bmin.y) o.y /= bmax.y - bmin.y;
        if (bmax.z > bmin.z) o.z /= bmax.z - bmin.z;
        int pi[3];
        for (int i = 0; i < 3; ++i) pi[i] = Clamp(int(o[i] * nVoxels[i]), 0, nVoxels[i] - 1);
        size_t idx = ((size_t)pi[2] * nVoxels[1] + pi[1]) * nVoxels[0] + pi[0];
        Distribution1D *dist = voxels[idx].load(std::memory_order_acquire);
        if (dist) return dist;
        std::lock_guard<std::mutex> lock(mu);
        dist = voxels[idx].load(std::memory_order_acquire);
        if (!dist) {
            dist = ComputeDistribution(pi);
            voxels[idx].store(dist, std::memory_order_release);
        }
        return dist;
    }
};

// ------------------------------------------------------------------ integrator
inline Float PowerHeuristic(int nf, Float fPdf, int ng, Float gPdf) {  // sampling.h:171-174
    Float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}

static Spec PrimLe(const mi_scene_desc &d, const SurfaceInteraction &isect, const V3 &w) {  // interaction.cpp:150-153
    int li = d.prims[isect.prim].area_light;
    if (li < 0) return Spec(0.f);
    const mi_light &l = d.lights[li];
    return (l.two_sided || Dot(isect.n, w) > 0) ? Spec::From(l.L) : Spec(0.f);
}

static Spec EstimateDirect(const Scene &scene, const SurfaceInteraction &it, const BSDF &bsdf, const Float uScattering[2],
                           int lightNum, const Float uLight[2], Counters &c) {  // integrator.cpp:108-215, specular=false
    const mi_scene_desc &d = scene.d;
    const mi_light &light = d.lights[lightNum];
    const int bsdfFlags = MI_BSDF_ALL & ~MI_BSDF_SPECULAR;
    Spec Ld(0.f);
    Float lightPdf = 0, scatteringPdf = 0;
    LightSample ls = SampleLi(d, light, it, uLight);
    lightPdf = ls.pdf;
    Spec Li = ls.Li;
    V3 wi = ls.wi;
    if (lightPdf > 0 && !Li.IsBlack()) {
        Spec f = bsdf.f(it.wo, wi, bsdfFlags) * AbsDot(wi, it.shading.n);
        scatteringPdf = bsdf.Pdf(it.wo, wi, bsdfFlags);
        if (!f.IsBlack()) {
            if (scene.IntersectP(SpawnRayTo(it, ls.pLight), c)) Li = Spec(0.f);
            if (!Li.IsBlack()) {
                if (IsDeltaLight(light)) Ld += f * Li / lightPdf;
                else {
                    Float weight = PowerHeuristic(1, lightPdf, 1, scatteringPdf);
                    Ld += f * Li * weight / lightPdf;
                }
            }
        }
    }
    if (!IsDeltaLight(light)) {
        int sampledType = 0;
        Spec f = bsdf.Sample_f(it.wo, &wi, uScattering, &scatteringPdf, bsdfFlags, &sampledType);
        f *= AbsDot(wi, it.shading.n);
        bool sampledSpecular = (sampledType & MI_BSDF_SPECULAR) != 0;
        if (!f.IsBlack() && scatteringPdf > 0) {
            Float weight = 1;
            if (!sampledSpecular) {
                lightPdf = (light.type == MI_LIGHT_INFINITE) ? InfinitePdfLi(d, light, wi)
                                                             : ShapePdf(d, light.shape, light.area, it, wi);  // DiffuseAreaLight::Pdf_Li
                if (lightPdf == 0) return Ld;
                weight = PowerHeuristic(1, scatteringPdf, 1, lightPdf);
            }
            SurfaceInteraction lightIsect;
            Ray ray = SpawnRay(it, wi);
            bool found = scene.Intersect(ray, &lightIsect, c);
            Spec Li2(0.f);
            if (found) {
                if (d.prims[lightIsect.prim].area_light == lightNum) Li2 = PrimLe(d, lightIsect, -wi);
            } else if (light.type == MI_LIGHT_INFINITE) Li2 = InfiniteLe(d, light, ray.d);   // light.Le(ray); 0 for the other types (light.cpp:86)
            if (!Li2.IsBlack()) Ld += f * Li2 * weight / scatteringPdf;
        }
    }
    return Ld;
}

// Optional per-vertex log of Li (oracle_path_log; record layout: include/mi_pt.h, MI_PATH_RECORD_FLOATS).
static thread_local std::vector<float> *g_pathLog = nullptr;
struct PathLogRecord {
    float *r = nullptr;
    void Open(int bounces, int prim, int dim, const Ray &ray, Float etaScale) {
        if (!g_pathLog) return;
        g_pathLog->resize(g_pathLog->size() + MI_PATH_RECORD_FLOATS, 0.f);
        r = g_pathLog->data() + g_pathLog->size() - MI_PATH_RECORD_FLOATS;
        r[0] = (float)bounces; r[1] = (float)prim; r[2] = (float)dim; r[3] = 1.f;
        r[4] = ray.o.x; r[5] = ray.o.y; r[6] = ray.o.z; r[7] = ray.tMax;
        r[8] = ray.d.x; r[9] = ray.d.y; r[10] = ray.d.z; r[11] = etaScale;
    }
    void Close(bool ended, const Ray &next, int dim, Float etaScale, const Spec &beta, const Spec &L) {
        if (!r) return;
        r[3] = ended ? 1.f : 0.f;
        r[12] = next.o.x; r[13] = next.o.y; r[14] = next.o.z; r[15] = (float)dim;
        r[16] = next.d.x; r[17] = next.d.y; r[18] = next.d.z; r[19] = etaScale;
        for (int k = 0; k < NS; ++k) { r[20 + k] = beta.c[k]; r[51 + k] = L.c[k]; }
    }
};

static Spec Li(const Scene &scene, LightDistribution &lightDistrib, const Ray &r, Sampler &sampler, Counters &c,
               const RayDifferential &camDiff = RayDifferential()) {
    // path.cpp:64-188
    const mi_scene_desc &d = scene.d;
    const int maxDepth = d.integrator.max_depth;
    const Float rrThreshold = d.integrator.rr_threshold;
    Spec L(0.f), beta(1.f);
    Ray ray(r);
    RayDifferential diff = camDiff;   // only the camera ray carries differentials; SpawnRay() returns a plain Ray
    bool specularBounce = false;
    int bounces;
    Float etaScale = 1;
    for (bounces = 0;; ++bounces) {
        SurfaceInteraction isect;
        bool foundIntersection = scene.Intersect(ray, &isect, c);
        PathLogRecord rec;
        rec.Open(bounces, foundIntersection ? isect.prim : -1, sampler.dimension, ray, etaScale);
        struct AtExit {   // whatever way the iteration ends, the record is closed with the state at that point
            PathLogRecord &rec; const Ray &ray; Sampler &sampler; Float &etaScale; Spec &beta, &L; bool ended = true;
            ~AtExit() { rec.Close(ended, ray, sampler.dimension, etaScale, beta, L); }
        } atExit{rec, ray, sampler, etaScale, beta, L};
        if (bounces == 0 || specularBounce) {
            if (foundIntersection) L += beta * PrimLe(d, isect, -ray.d);
            else
                for (uint32_t i = 0; i < d.n_lights; ++i)   // scene.infiniteLights, path.cpp:96-99
                    if (d.lights[i].type == MI_LIGHT_INFINITE) L += beta * InfiniteLe(d, d.lights[i], ray.d);
        }
        if (!foundIntersection || bounces >= maxDepth) break;
        int matIdx = d.prims[isect.prim].material;
        if (matIdx < 0) {  // !isect.bsdf, path.cpp:108-113
            ray = SpawnRay(isect, ray.d);
            diff.hasDifferentials = false;
            bounces--;
            atExit.ended = false;
            continue;
        }
        const mi_material &mat = d.materials[matIdx];
        TexDifferentials td;
        if (mat.textured) td = ComputeDifferentials(isect, diff);   // isect.ComputeScatteringFunctions(ray, ...), interaction.cpp:91-97
        if (mat.bump_tex >= 0) Bump(d, mat.bump_tex, &isect, td);
        BSDF bsdf(isect, mat, &d, &td);
        const Distribution1D *distrib = lightDistrib.Lookup(isect.p);
        if (bsdf.NumComponents(MI_BSDF_ALL & ~MI_BSDF_SPECULAR) > 0) {
            ++c.totalPaths;
            // UniformSampleOneLight, integrator.cpp:85-106
            Spec Ld(0.f);
            int nLights = (int)d.n_lights;
            if (nLights > 0) {
                Float lightPdf;
                int lightNum = distrib->SampleDiscrete(sampler.Get1D(), &lightPdf);
                if (lightPdf != 0) {
                    Float uLight[2], uScattering[2];
                    sampler.Get2D(uLight);
                    sampler.Get2D(uScattering);
                    Ld = beta * (EstimateDirect(scene, isect, bsdf, uScattering, lightNum, uLight, c) / lightPdf);
                }
            }
            if (Ld.IsBlack()) ++c.zeroRadiancePaths;
            L += Ld;
        }
        V3 wo = -ray.d, wi;
        Float pdf = 0;
        int flags = 0;
        Float u2[2];
        sampler.Get2D(u2);
        Spec f = bsdf.Sample_f(wo, &wi, u2, &pdf, MI_BSDF_ALL, &flags);
        if (f.IsBlack() || pdf == 0.f) break;
        beta *= f * AbsDot(wi, isect.shading.n) / pdf;
        specularBounce = (flags & MI_BSDF_SPECULAR) != 0;
        if ((flags & MI_BSDF_SPECULAR) && (flags & MI_BSDF_TRANSMISSION)) {
            Float eta = bsdf.eta;
            etaScale *= (Dot(wo, isect.n) > 0) ? (eta * eta) : 1 / (eta * eta);
        }
        ray = SpawnRay(isect, wi);
        diff.hasDifferentials = false;
        Spec rrBeta = beta * etaScale;
        if (rrBeta.MaxComponentValue() < rrThreshold && bounces > 3) {
            Float q = std::max((Float).05, 1 - rrBeta.MaxComponentValue());
            if (sampler.Get1D() < q) break;
            beta /= 1 - q;
        }
        atExit.ended = false;
    }
    c.pathLengthSum += bounces;
    return L;
}

// ------------------------------------------------------------------ camera
static Ray GenerateRay(const mi_scene_desc &d, const Float pFilm[2], const Float pLens[2], RayDifferential *rd = nullptr) {
    // PerspectiveCamera::GenerateRayDifferential, perspective.cpp:95-146
    const mi_camera &cam = d.camera;
    V3 pCamera = XfPoint(cam.raster_to_camera, V3(pFilm[0], pFilm[1], 0));
    V3 dir = Normalize(V3(pCamera.x, pCamera.y, pCamera.z));
    Ray ray(V3(0, 0, 0), dir);
    if (cam.lens_radius > 0) {
        Float dl[2];
        ConcentricSampleDisk(pLens, dl);
        Float lx = cam.lens_radius * dl[0], ly = cam.lens_radius * dl[1];
        Float ft = cam.focal_distance / ray.d.z;
        V3 pFocus = ray(ft);
        ray.o = V3(lx, ly, 0);
        ray.d = Normalize(pFocus - ray.o);
    }
    if (rd) {
        // dxCamera, dyCamera: perspective.cpp:54-58
        const V3 r0 = XfPoint(cam.raster_to_camera, V3(0, 0, 0));
        const V3 dxCamera = XfPoint(cam.raster_to_camera, V3(1, 0, 0)) - r0, dyCamera = XfPoint(cam.raster_to_camera, V3(0, 1, 0)) - r0;
        V3 rxO, ryO, rxD, ryD;
        if (cam.lens_radius > 0) {
            Float dl[2];
            ConcentricSampleDisk(pLens, dl);
            Float lx = cam.lens_radius * dl[0], ly = cam.lens_radius * dl[1];
            V3 dx = Normalize(V3(pCamera + dxCamera));
            Float ft = cam.focal_distance / dx.z;
            V3 pFocus = V3(0, 0, 0) + (ft * dx);
            rxO = V3(lx, ly, 0);
            rxD = Normalize(pFocus - rxO);
            V3 dy = Normalize(V3(pCamera + dyCamera));
            ft = cam.focal_distance / dy.z;
            pFocus = V3(0, 0, 0) + (ft * dy);
            ryO = V3(lx, ly, 0);
            ryD = Normalize(pFocus - ryO);
        } else {
            rxO = ryO = ray.o;
            rxD = Normalize(V3(pCamera) + dxCamera);
            ryD = Normalize(V3(pCamera) + dyCamera);
        }
        // Transform::operator()(RayDifferential), transform.h:268-277
        rd->rxOrigin = XfPoint(cam.camera_to_world, rxO);
        rd->ryOrigin = XfPoint(cam.camera_to_world, ryO);
        rd->rxDirection = XfVector(cam.camera_to_world, rxD);
        rd->ryDirection = XfVector(cam.camera_to_world, ryD);
        rd->hasDifferentials = true;
    }
    return XfRay(cam.camera_to_world, ray);
}

// ------------------------------------------------------------------ film
struct FilmTile {
    int x0, y0, x1, y1;
    std::vector<Spec> contribSum;
    std::vector<Float> filterWeightSum;
    FilmTile(int x0, int y0, int x1, int y1) : x0(x0), y0(y0), x1(x1), y1(y1) {
        size_t n = (size_t)std::max(0, (x1 - x0) * (y1 - y0));
        if (x1 <= x0 || y1 <= y0) n = 0;
        contribSum.resize(n);
        filterWeightSum.assign(n, 0.f);
    }
    void AddSample(const mi_scene_desc &d, const Float pFilm[2], Spec L, Float sampleWeight) {  // film.h:123-163
        const mi_film &f = d.film;
        if (SpecY(d, L) > f.max_sample_luminance) L *= f.max_sample_luminance / SpecY(d, L);
        const int filterTableSize = 16;
        Float dx = pFilm[0] - 0.5f, dy = pFilm[1] - 0.5f;
        int p0x = (int)std::ceil(dx - f.filter_radius[0]), p0y = (int)std::ceil(dy - f.filter_radius[1]);
        int p1x = (int)std::floor(dx + f.filter_radius[0]) + 1, p1y = (int)std::floor(dy + f.filter_radius[1]) + 1;
        p0x = std::max(p0x, x0); p0y = std::max(p0y, y0);
        p1x = std::min(p1x, x1); p1y = std::min(p1y, y1);
        Float invRx = 1 / f.filter_radius[0], invRy = 1 / f.filter_radius[1];
        for (int y = p0y; y < p1y; ++y) {
            Float fy = std::abs((y - dy) * invRy * filterTableSize);
            int iy = std::min((int)std::floor(fy), filterTableSize - 1);
            for (int x = p0x; x < p1x; ++x) {
                Float fx = std::abs((x - dx) * invRx * filterTableSize);
                int ix = std::min((int)std::floor(fx), filterTableSize - 1);
                Float filterWeight = f.filter_table[iy * filterTableSize + ix];
                size_t off = (size_t)(x - x0) + (size_t)(y - y0) * (x1 - x0);
                contribSum[off] += L * sampleWeight * filterWeight;
                filterWeightSum[off] += filterWeight;
            }
        }
    }
};

struct RenderJob {
    const mi_scene_desc &d;
    Scene scene;
    LightDistribution lightDistrib;
    int shardIndex, shardCount;
    int64_t spp;
    int nTilesX, nTilesY;
    std::atomic<int> nextTile{0};
    std::mutex filmMutex;
    float *filmSum, *weightSum;  // cropped-bounds pixel-major
    Counters total;
    int64_t maxSamples;  // <0: all; otherwise stop handing out tiles once this many samples are done (bench)
    std::atomic<int64_t> samplesDone{0};

    RenderJob(const mi_scene_desc &d, int si, int sc, float *film, float *weight)
        : d(d), scene(d), lightDistrib(d), shardIndex(si), shardCount(sc), filmSum(film), weightSum(weight) {
        spp = d.sampler.samples_per_pixel;
        const int *sb = d.film.sample_bounds;
        nTilesX = (sb[2] - sb[0] + 15) / 16;
        nTilesY = (sb[3] - sb[1] + 15) / 16;
        maxSamples = -1;
    }

    void RenderTile(int tile, Counters &c) {
        const mi_film &f = d.film;
        const int *sb = f.sample_bounds;
        const int tileSize = 16;
        int tx = tile % nTilesX, ty = tile / nTilesX;
        int x0 = sb[0] + tx * tileSize, x1 = std::min(x0 + tileSize, sb[2]);
        int y0 = sb[1] + ty * tileSize, y1 = std::min(y0 + tileSize, sb[3]);
        // GetFilmTile, film.cpp:101-112
        int p0x = (int)std::ceil((Float)x0 - 0.5f - f.filter_radius[0]);
        int p0y = (int)std::ceil((Float)y0 - 0.5f - f.filter_radius[1]);
        int p1x = (int)std::floor((Float)x1 - 0.5f + f.filter_radius[0]) + 1;
        int p1y = (int)std::floor((Float)y1 - 0.5f + f.filter_radius[1]) + 1;
        const int *cb = f.cropped_bounds;
        FilmTile ft(std::max(p0x, cb[0]), std::max(p0y, cb[1]), std::min(p1x, cb[2]), std::min(p1y, cb[3]));
        Sampler sampler(d);
        const int *pb = d.integrator.pixel_bounds;
        for (int py = y0; py < y1; ++py)
            for (int px = x0; px < x1; ++px) {
                sampler.StartPixel(px, py);
                if (!(px >= pb[0] && px < pb[2] && py >= pb[1] && py < pb[3])) continue;
                for (int64_t s = 0; s < spp; ++s) {
                    sampler.StartSample(s);
                    Float u[2], pLens[2];
                    sampler.Get2D(u);
                    Float pFilm[2] = {(Float)px + u[0], (Float)py + u[1]};
                    (void)sampler.Get1D();  // time
                    sampler.Get2D(pLens);
                    // SamplerIntegrator::Render (integrator.cpp:264-328) is the nBands == 1 case of
                    // SpectralPathIntegrator::Render (spectralpath.cpp:258-318): one path per band from
                    // the same camera sample, the sampler's dimension running on, band s supplying the
                    // bins [deltaIndex*s, min(deltaIndex*(s+1), nSpectralSamples)).
                    const int nBands = std::max(1, (int)d.integrator.n_ca_bands);
                    const int deltaIndex = (int)std::round((float)NS / (float)nBands);
                    Spec L(0.f);
                    for (int band = 0; band < nBands; ++band) {
                        RayDifferential rd;
                        Ray ray = GenerateRay(d, pFilm, pLens, &rd);
                        rd.ScaleDifferentials(ray.o, ray.d, 1 / std::sqrt((Float)d.sampler.samples_per_pixel));  // integrator.cpp:286-287
                        ++c.cameraRays;
                        Spec Ls = Li(scene, lightDistrib, ray, sampler, c, rd);
                        if (Ls.HasNaNs()) { Ls = Spec(0.f); ++c.badSamples; }
                        else if (SpecY(d, Ls) < -1e-5) { Ls = Spec(0.f); ++c.badSamples; }
                        else if (std::isinf(SpecY(d, Ls))) { Ls = Spec(0.f); ++c.badSamples; }
                        const int lo = deltaIndex * band, hi = std::min(deltaIndex * (band + 1), NS);
                        for (int k = lo; k < hi; ++k) L.c[k] = Ls.c[k];
                    }
                    ft.AddSample(d, pFilm, L, 1.f);
                }
            }
        // MergeFilmTile, film.cpp:124-142
        std::lock_guard<std::mutex> lock(filmMutex);
        int w = cb[2] - cb[0];
        for (int y = ft.y0; y < ft.y1; ++y)
            for (int x = ft.x0; x < ft.x1; ++x) {
                size_t toff = (size_t)(x - ft.x0) + (size_t)(y - ft.y0) * (ft.x1 - ft.x0);
                size_t poff = (size_t)(x - cb[0]) + (size_t)(y - cb[1]) * w;
                if (filmSum) for (int k = 0; k < NS; ++k) filmSum[poff * NS + k] += ft.contribSum[toff].c[k];
                if (weightSum) weightSum[poff] += ft.filterWeightSum[toff];
            }
    }

    void Worker() {
        Counters c;
        int nTiles = nTilesX * nTilesY;
        while (true) {
            if (maxSamples >= 0 && samplesDone.load() >= maxSamples) break;
            int t = nextTile.fetch_add(1);
            if (t >= nTiles) break;
            if (t % shardCount != shardIndex) continue;
            uint64_t before = c.cameraRays;
            RenderTile(t, c);
            samplesDone.fetch_add((int64_t)(c.cameraRays - before));
        }
        std::lock_guard<std::mutex> lock(filmMutex);
        total.Add(c);
    }
};

static void FillCounters(const Counters &t, mi_counters *out) {
    if (!out) return;
    *out = mi_counters{};
    out->camera_rays = t.cameraRays; out->regular_rays = t.regularRays; out->shadow_rays = t.shadowRays;
    out->total_paths = t.totalPaths; out->zero_radiance_paths = t.zeroRadiancePaths;
    out->path_length_sum = t.pathLengthSum; out->bvh_nodes_visited = t.nodesVisited; out->tri_tests = t.triTests;
    out->bad_samples = t.badSamples;
}

}  // namespace orc

using namespace orc;

extern "C" {

// Render the whole film (or the tiles of one shard). film_sum [H*W*31] and weight_sum
// [H*W] (either may be NULL) are ACCUMULATED into (caller zeroes them). max_samples<0
// renders everything; otherwise workers stop taking tiles after that many camera
// samples (bounded CPU-baseline timing). Returns wall seconds of the render loop.
double oracle_render(const mi_scene_desc *desc, int n_threads, int shard_index, int shard_count,
                     int64_t max_samples, float *film_sum, float *weight_sum, mi_counters *counters) {
    RenderJob job(*desc, shard_index, std::max(1, shard_count), film_sum, weight_sum);
    job.maxSamples = max_samples;
    auto t0 = std::chrono::steady_clock::now();
    int nt = std::max(1, n_threads);
    std::vector<std::thread> threads;
    for (int i = 1; i < nt; ++i) threads.emplace_back([&job] { job.Worker(); });
    job.Worker();
    for (auto &t : threads) t.join();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    FillCounters(job.total, counters);
    return secs;
}

// Radiance of single camera samples: samples = n x {px, py, sampleNum}; out = n x 31.
void oracle_li(const mi_scene_desc *desc, const int32_t *samples, int n, float *out, mi_counters *counters) {
    Scene scene(*desc);
    LightDistribution ld(*desc);
    Counters c;
    Sampler sampler(*desc);
    for (int i = 0; i < n; ++i) {
        int px = samples[3 * i], py = samples[3 * i + 1];
        sampler.StartPixel(px, py);
        sampler.StartSample(samples[3 * i + 2]);
        Float u[2], pLens[2];
        sampler.Get2D(u);
        Float pFilm[2] = {(Float)px + u[0], (Float)py + u[1]};
        (void)sampler.Get1D();
        sampler.Get2D(pLens);
        RayDifferential rd;
        Ray ray = GenerateRay(*desc, pFilm, pLens, &rd);
        rd.ScaleDifferentials(ray.o, ray.d, 1 / std::sqrt((Float)desc->sampler.samples_per_pixel));
        ++c.cameraRays;
        Spec L = Li(scene, ld, ray, sampler, c, rd);
        for (int k = 0; k < NS; ++k) out[(size_t)i * NS + k] = L.c[k];
    }
    FillCounters(c, counters);
}

// Vertex-by-vertex log of one camera sample (record layout: include/mi_pt.h, MI_PATH_RECORD_FLOATS). Returns the number of records.
int oracle_path_log(const mi_scene_desc *desc, int px, int py, int64_t sample, int max_records, float *records) {
    std::vector<float> log;
    g_pathLog = &log;
    const int32_t smp[3] = {px, py, (int32_t)sample};
    float out[NS];
    mi_counters c;
    oracle_li(desc, smp, 1, out, &c);
    g_pathLog = nullptr;
    const int n = std::min<int>(max_records, (int)(log.size() / MI_PATH_RECORD_FLOATS));
    memcpy(records, log.data(), (size_t)n * MI_PATH_RECORD_FLOATS * sizeof(float));
    return n;
}

// Camera rays for given samples: out = n x {o[3], d[3], tMax}.
void oracle_camera_rays(const mi_scene_desc *desc, const int32_t *samples, int n, float *out) {
    Sampler sampler(*desc);
    for (int i = 0; i < n; ++i) {
        int px = samples[3 * i], py = samples[3 * i + 1];
        sampler.StartPixel(px, py);
        sampler.StartSample(samples[3 * i + 2]);
        Float u[2], pLens[2];
        sampler.Get2D(u);
        Float pFilm[2] = {(Float)px + u[0], (Float)py + u[1]};
        (void)sampler.Get1D();
        sampler.Get2D(pLens);
        Ray ray = GenerateRay(*desc, pFilm, pLens);
        float *o = out + (size_t)i * 7;
        o[0] = ray.o.x; o[1] = ray.o.y; o[2] = ray.o.z; o[3] = ray.d.x; o[4] = ray.d.y; o[5] = ray.d.z; o[6] = ray.tMax;
    }
}

// BVH traversal on recorded rays (same layout as mi_pt_trace): rays n x 7, hits n x 4
// {prim as int bits (-1 miss), t, b0, b1}; for spheres b0/b1 are 0.
void oracle_trace(const mi_scene_desc *desc, const float *rays, uint32_t n, int any_hit, float *hits,
                  mi_counters *counters) {
    Scene scene(*desc);
    Counters c;
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = rays + (size_t)i * 7;
        Ray ray(V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), r[6]);
        float *h = hits + (size_t)i * 4;
        int32_t prim = -1;
        h[1] = h[2] = h[3] = 0;
        if (any_hit) {
            prim = scene.IntersectP(ray, c) ? 0 : -1;
        } else {
            SurfaceInteraction isect;
            if (scene.Intersect(ray, &isect, c)) {
                prim = isect.prim;
                h[1] = ray.tMax;
                const mi_prim &p = desc->prims[prim];
                if (p.shape >= 0) {
                    TriVerts tv = GetTri(*desc, p.shape);
                    Ray r2(ray.o, ray.d, r[6]);
                    TriHit th;
                    if (TriTest(tv.p0, tv.p1, tv.p2, r2, &th)) { h[2] = th.b0; h[3] = th.b1; }
                }
            }
        }
        memcpy(&h[0], &prim, 4);
    }
    FillCounters(c, counters);
}

// libm evaluation mode of the oracle (o_math.h): 0 = the host's float functions, 1 = correctly rounded. Returns the previous mode.
int oracle_set_libm(int mode) { const int old = g_libmMode; g_libmMode = mode ? 1 : 0; return old; }

// ---- unit-level entry points for pinning against the reference's own tests
float oracle_radical_inverse(const mi_scene_desc *desc, int base_index, uint64_t a) { return RadicalInverse(*desc, base_index, a); }
float oracle_scrambled_radical_inverse(const mi_scene_desc *desc, int base_index, uint64_t a) {
    const mi_sampler &s = desc->sampler;
    return ScrambledRadicalInverseBase(s.primes[base_index], &s.perms[s.prime_sums[base_index]], a);
}
float oracle_sample_dimension(const mi_scene_desc *desc, int px, int py, int64_t sample_num, int dim) {
    Sampler s(*desc);
    s.StartPixel(px, py);
    s.StartSample(sample_num);
    return s.SampleDimension(s.intervalSampleIndex, dim);
}
// Scalar helpers on their own, for the reference's FloatingPoint.NextUpDownFloat, EFloat.{Add,Sub,Mul,Div} and
// FindInterval.Basics tests (tests/fp_tests.cpp:29-47,166-260, tests/find_interval.cpp:8) -- the same operations and layout as
// the device's mi_pt_math_probe. x, y: 2 floats per element, out: 3 floats per element.
//   op 0: NextFloatUp(x0), NextFloatDown(x0)      op 1..4: EFloat(x0, x1) {+, -, *, /} EFloat(y0, y1) -> v, LowerBound, UpperBound
//   op 5: FindInterval(10, [&](int i) { return i <= x0; }) over the array 0, 1, ..., 9, as Distribution1D::SampleDiscrete runs it
void oracle_math_probe(int op, uint32_t n, const float *x, const float *y, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1], y0 = y ? y[2 * i] : 0.f, y1 = y ? y[2 * i + 1] : 0.f;
        float *o = out + 3 * i;
        o[0] = o[1] = o[2] = 0.f;
        if (op == 0) { o[0] = NextFloatUp(x0); o[1] = NextFloatDown(x0); }
        else if (op >= 1 && op <= 4) {
            const EFloat a(x0, x1), b(y0, y1);
            const EFloat r = op == 1 ? a + b : (op == 2 ? a - b : (op == 3 ? a * b : a / b));
            o[0] = r.v; o[1] = r.low; o[2] = r.high;
        } else if (op == 5) {
            const float cdf[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9}, func[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
            Distribution1D dist(func, 9);
            dist.cdf.assign(cdf, cdf + 10);   // (the reference test's array in place of a cdf: FindInterval sees the same values)
            float pdf;
            o[0] = (float)dist.SampleDiscrete(x0, &pdf);
        }
    }
}
// The pieces the reference's LowDiscrepancy.GeneratorMatrix / GrayCodeSample tests exercise (tests/sampling.cpp:76-118).
uint32_t oracle_multiply_generator(const uint32_t *C, uint32_t a) { return MultiplyGenerator(C, a); }
float oracle_sample_generator_matrix(const uint32_t *C, uint32_t a, uint32_t scramble) { return SampleGeneratorMatrix(C, a, scramble); }
void oracle_gray_code_sample(const uint32_t *C, uint32_t n, uint32_t scramble, float *out) { GrayCodeSample(C, n, scramble, out); }
void oracle_zerotwo_matrices(uint32_t *vdc32, uint32_t *sobol32) { memcpy(vdc32, kZeroTwo.vdc, 128); memcpy(sobol32, kZeroTwo.sobol1, 128); }
// One sample's first calls through the sampler interface: out = {Get2D (2), Get1D, Get2D (2), Get1D, Get2D (2), ...} n values
// alternating one 2D and one 1D call -- the camera sample's order (sampler.cpp:46-52) continued.
void oracle_sampler_calls(const mi_scene_desc *desc, int px, int py, int64_t sample_num, int n_pairs, float *out) {
    Sampler s(*desc);
    s.StartPixel(px, py);
    s.StartSample(sample_num);
    for (int i = 0; i < n_pairs; ++i) { s.Get2D(out + 3 * i); out[3 * i + 2] = s.Get1D(); }
}
// SobolSampleFloat(index, dim) itself (lowdiscrepancy.h:259-274), for the reference's LowDiscrepancy.Sobol test.
float oracle_sobol_sample(const mi_scene_desc *desc, int64_t index, int dim) { return SobolSampleFloat(desc->sampler, index, dim); }
// Single triangle test (tests/shapes.cpp Triangle.*): p = 9 floats, ray = 7 floats.
int oracle_tri_test(const float *p, const float *ray, float *out4) {
    Ray r(V3(ray[0], ray[1], ray[2]), V3(ray[3], ray[4], ray[5]), ray[6]);
    TriHit h;
    if (!TriTest(V3(p[0], p[1], p[2]), V3(p[3], p[4], p[5]), V3(p[6], p[7], p[8]), r, &h)) return 0;
    out4[0] = h.t; out4[1] = h.b0; out4[2] = h.b1; out4[3] = h.b2;
    return 1;
}
// BSDF of material `mat` in a canonical frame (n = +z, dpdu = +x): mode 0 -> f and pdf
// for (wo, wi); mode 1 -> Sample_f(wo, u). out: f[31], pdf, wi[3], flags.
void oracle_bsdf(const mi_scene_desc *desc, int mat, int mode, const float *wo3, const float *wi3, const float *u2,
                 int flags, float *out) {
    SurfaceInteraction si;
    si.n = si.shading.n = V3(0, 0, 1);
    si.dpdu = si.shading.dpdu = V3(1, 0, 0);
    si.dpdv = si.shading.dpdv = V3(0, 1, 0);
    BSDF bsdf(si, desc->materials[mat]);
    V3 wo(wo3[0], wo3[1], wo3[2]);
    if (mode == 0) {
        V3 wi(wi3[0], wi3[1], wi3[2]);
        Spec f = bsdf.f(wo, wi, flags);
        for (int k = 0; k < NS; ++k) out[k] = f.c[k];
        out[31] = bsdf.Pdf(wo, wi, flags);
        out[32] = wi.x; out[33] = wi.y; out[34] = wi.z; out[35] = 0;
    } else {
        V3 wi;
        Float pdf = 0;
        int st = 0;
        Float u[2] = {u2[0], u2[1]};
        Spec f = bsdf.Sample_f(wo, &wi, u, &pdf, flags, &st);
        for (int k = 0; k < NS; ++k) out[k] = f.c[k];
        out[31] = pdf; out[32] = wi.x; out[33] = wi.y; out[34] = wi.z; out[35] = (float)st;
    }
}
// Spatial light distribution of the voxel containing p: pmf[n_lights].
void oracle_light_pmf(const mi_scene_desc *desc, const float *p3, float *pmf) {
    LightDistribution ld(*desc);
    const Distribution1D *dist = ld.Lookup(V3(p3[0], p3[1], p3[2]));
    for (uint32_t i = 0; dist && i < desc->n_lights; ++i)
        pmf[i] = (dist->funcInt > 0) ? dist->func[i] / (dist->funcInt * dist->Count()) : 0;
}

// Shape::Sample(ref, u, &pdf) (shape.cpp:56-70, sphere.cpp:232-292, triangle.cpp:583-608) of shape `shape`
// (>= 0 triangle, < 0 ~sphere) for a reference point as the reference's tests build it
// (tests/shapes.cpp:240: Interaction ref(pc, Normal3f(), Vector3f(), Vector3f(0, 0, 1), ...): no normal, no error bound).
// n samples u2[n][2] -> out7[n] = {p.xyz, n.xyz, pdf}.
void oracle_shape_sample(const mi_scene_desc *desc, int shape, const float *ref3, int n, const float *u2, float *out7) {
    Interaction ref;
    ref.p = V3(ref3[0], ref3[1], ref3[2]);
    ref.pError = V3(0, 0, 0); ref.n = V3(0, 0, 0); ref.wo = V3(0, 0, 1);
    for (int i = 0; i < n; ++i) {
        const Float u[2] = {u2[2 * i], u2[2 * i + 1]};
        Float pdf = 0;
        Interaction it = ShapeSample(*desc, shape, ref, u, &pdf);
        float *o = out7 + (size_t)i * 7;
        o[0] = it.p.x; o[1] = it.p.y; o[2] = it.p.z;
        o[3] = it.n.x; o[4] = it.n.y; o[5] = it.n.z;
        o[6] = pdf;
    }
}
// Shape::Pdf(ref, wi) (shape.cpp:72-87, sphere.cpp:294-306) for the same kind of reference point.
float oracle_shape_pdf(const mi_scene_desc *desc, int shape, float area, const float *ref3, const float *wi3) {
    Interaction ref;
    ref.p = V3(ref3[0], ref3[1], ref3[2]);
    ref.pError = V3(0, 0, 0); ref.n = V3(0, 0, 0); ref.wo = V3(0, 0, 1);
    return ShapePdf(*desc, shape, area, ref, V3(wi3[0], wi3[1], wi3[2]));
}
// The rays a SurfaceInteraction spawns (tests/shapes.cpp:154-205, 375-425): intersect `ray7` with the scene; at the hit,
// for each of the n `targets` (mode 0: a direction w -> isect.SpawnRay(w); mode 1: a point p2 -> isect.SpawnRayTo(p2)),
// optionally flipped into the hemisphere of the surface normal first (Faceforward, the convex-shape variant), write the
// spawned ray {o, d, tMax} to out_rays. Returns 1 if the first ray hit, else 0 (then nothing is written).
int oracle_spawn_rays(const mi_scene_desc *desc, const float *ray7, int n, const float *targets, int mode, int faceforward,
                      float *out_rays) {
    Scene scene(*desc);
    Counters c;
    Ray ray(V3(ray7[0], ray7[1], ray7[2]), V3(ray7[3], ray7[4], ray7[5]), ray7[6]);
    SurfaceInteraction isect;
    if (!scene.Intersect(ray, &isect, c)) return 0;
    for (int i = 0; i < n; ++i) {
        V3 t(targets[3 * i], targets[3 * i + 1], targets[3 * i + 2]);
        Ray r;
        if (mode == 0) {
            if (faceforward) t = Faceforward(t, isect.n);
            r = SpawnRay(isect, t);
        } else {
            if (faceforward) {
                V3 w = t - isect.p;
                w = Faceforward(w, isect.n);
                t = isect.p + w;
            }
            r = Ray(OffsetRayOrigin(isect.p, isect.pError, isect.n, t - isect.p), t - isect.p, 1 - ShadowEpsilon);
        }
        float *o = out_rays + (size_t)i * 7;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.tMax;
    }
    return 1;
}

// The raw table of voxel (x, y, z): Distribution1D::func [n_lights] and funcInt (lightdistrib.cpp:232-300).
void oracle_light_voxel(const mi_scene_desc *desc, const int *pi3, float *func, float *func_int) {
    LightDistribution ld(*desc);
    if (desc->n_lights == 0 || desc->light_distrib.type != MI_LD_SPATIAL) return;
    std::unique_ptr<Distribution1D> dist(ld.ComputeDistribution(pi3));
    for (uint32_t i = 0; i < desc->n_lights; ++i) func[i] = dist->func[i];
    *func_int = dist->funcInt;
}

// ... and of every voxel at once (index (z * ny + y) * nx + x), on n_threads threads: func [n_voxels * n_lights], func_int [n_voxels].
void oracle_light_table(const mi_scene_desc *desc, int n_threads, float *func, float *func_int) {
    if (desc->n_lights == 0 || desc->light_distrib.type != MI_LD_SPATIAL) return;
    LightDistribution ld(*desc);
    const int *nv = desc->light_distrib.n_voxels;
    const size_t nVox = (size_t)nv[0] * nv[1] * nv[2];
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (;;) {
            const size_t v = next.fetch_add(1);
            if (v >= nVox) return;
            const int pi[3] = {(int)(v % nv[0]), (int)((v / nv[0]) % nv[1]), (int)(v / ((size_t)nv[0] * nv[1]))};
            std::unique_ptr<Distribution1D> dist(ld.ComputeDistribution(pi));
            for (uint32_t i = 0; i < desc->n_lights; ++i) func[v * desc->n_lights + i] = dist->func[i];
            func_int[v] = dist->funcInt;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < std::max(1, n_threads); ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

// Distribution1D over func[n] (sampling.h:55-109): mode 0 = SampleDiscrete(u) -> out{offset, pdf};
// mode 1 = SampleContinuous(u) -> out{x, pdf, offset}; mode 2 = DiscretePDF(index = (int)u) -> out{pdf}.
void oracle_distribution1d(const float *func, int n, float u, int mode, float *out) {
    Distribution1D dist(func, n);
    if (mode == 0) {
        Float pdf;
        out[0] = (float)dist.SampleDiscrete(u, &pdf);
        out[1] = pdf;
    } else if (mode == 1) {
        Float pdf;
        int off;
        out[0] = SampleContinuous1D(dist.func.data(), dist.cdf.data(), dist.funcInt, n, u, &pdf, &off);
        out[1] = pdf;
        out[2] = (float)off;
    } else {
        const int index = (int)u;
        out[0] = dist.func[index] / (dist.funcInt * dist.Count());
    }
}

// MIPMap<RGBSpectrum>::Lookup(st, dstdx, dstdy) of image texture `tex` (mipmap.h:281-319), and the spectrum
// ImageTexture::Evaluate makes of it (imagemap.h:82-93): out[0..2] = RGB, out[3..33] = Spectrum::FromRGB(rgb).
void oracle_texture_lookup(const mi_scene_desc *desc, int tex, const float *st2, const float *d4, float *out34) {
    const mi_texture &t = desc->textures[tex];
    for (int i = 0; i < 34; ++i) out34[i] = 0;
    if (t.type == MI_TEX_CHECKERBOARD) {   // out[0] = weight of tex2, out[3..33] = the spectrum
        const Float st[2] = {st2[0], st2[1]}, dx[2] = {d4[0], d4[1]}, dy[2] = {d4[2], d4[3]};
        const Float area2 = CheckerboardArea2(st, dx, dy, t.aa_none != 0);
        out34[0] = area2;
        const Spec sp = (1 - area2) * Spec::From(t.spec1) + area2 * Spec::From(t.spec2);
        for (int i = 0; i < NS; ++i) out34[3 + i] = sp.c[i];
        return;
    }
    if (t.type != MI_TEX_IMAGEMAP) return;
    MipView mip{desc->mipmaps[t.mipmap]};
    const Float st[2] = {st2[0], st2[1]}, dx[2] = {d4[0], d4[1]}, dy[2] = {d4[2], d4[3]};
    const RGB3 v = mip.Lookup(st, dx, dy, t.filter, t.max_aniso);
    for (int i = 0; i < 3; ++i) out34[i] = v.c[i];
    const Spec sp = SpecFromRGBIllum(*desc, v.c);
    for (int i = 0; i < NS; ++i) out34[3 + i] = sp.c[i];
}

}  // extern "C"
