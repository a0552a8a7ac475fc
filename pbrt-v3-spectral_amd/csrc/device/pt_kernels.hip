// pt_kernels.hip -- wavefront (Laine-style) spectral path tracer for gfx950 + the
// mi_pt_* C ABI (include/mi_pt.h). One resident pool of path slots lives in HBM as
// structure-of-arrays "planes" (plane k of slot i at base + k*pool + i, so every
// per-bin access of a wave is one coalesced 256-B transaction). One iteration is
//
//   generate : flush finished paths into the film, refill empty slots from a global
//              work counter (pixel, sample#) -> Halton camera sample -> camera ray
//   extend   : BVH2 closest-hit traversal, short stack staged in LDS
//   shade    : emission, termination, BSDF frame, light choice + light sample (NEE),
//              BSDF sample for MIS, BSDF sample for the continuation, Russian roulette;
//              streams the 31 bins of beta / contributions through HBM planes
//   shadow   : any-hit traversal of the NEE shadow rays, adds unoccluded contributions
//   mis      : the BSDF-sampled rays of the direct-lighting estimates: is the closest hit the
//              sampled light's shape? (a visibility query up to that shape, k_trav MODE 3; the
//              closest-hit traversal, MODE 2, for scenes with instances or a masked emitter); adds the
//              emission if so; closes the per-vertex direct-lighting statistics
//
// which restates, per path vertex, PathIntegrator::Li (src/integrators/path.cpp:64-188)
// with UniformSampleOneLight / EstimateDirect (src/core/integrator.cpp:85-215) inside
// SamplerIntegrator::Render's sample loop (src/core/integrator.cpp:228-342). Halton
// dimensions are consumed in the reference's order, so every path makes the same
// decisions as the CPU reference up to floating-point differences of libm functions.
// No MFMA: this path is latency/bandwidth bound (traversal) and VALU bound (shading).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cstring>
#include <string>
#include <vector>
#include <type_traits>
#include "d_sampling.h"
#include "d_texture.h"
#ifdef MIPT_SORT_EXPERIMENT
#include <hipcub/hipcub.hpp>
#endif

using namespace dpt;

// The library is built from this file four times in parallel (Makefile): MIPT_PART 0 holds everything but the shading
// kernel's instances (host code, the other kernels), parts 1-3 hold a third of the k_shade instances each (they are 95 % of
// the compile time). Without MIPT_PART (tools/isa_stats.py, tools/build_variants.sh) everything is one translation unit.
#ifdef MIPT_PART
#define MIPT_HAS_MAIN (MIPT_PART == 0)
#else
#define MIPT_HAS_MAIN 1
#endif

namespace dptk {   // (named: k_shade's instances are shared between the parts, so the types in its signature need linkage)

static thread_local std::string g_err;

// Four sub-renderer streams want four hardware queues of their own; the HIP runtime maps
// streams onto GPU_MAX_HW_QUEUES (default 4, shared with the null stream) when it
// initialises, which happens at the first HIP call -- after this library is loaded.
#if MIPT_HAS_MAIN
static struct QueueEnv { QueueEnv() { setenv("GPU_MAX_HW_QUEUES", "8", 0); } } g_queueEnv;
#endif
#define HIPCHK(x)                                                                         \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            g_err = std::string(#x) + ": " + hipGetErrorString(e_);                       \
            return MI_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

constexpr int BLOCK = 256;
#ifndef MIPT_STACK_LDS
#define MIPT_STACK_LDS 12
#endif
constexpr int STACK_LDS = MIPT_STACK_LDS;       // traversal stack entries (node, meta, tMin) staged in LDS per lane
constexpr int STACK_SPILL = 64 - STACK_LDS;     // deeper entries (pbrt allows 64 in total, bvh.cpp:670)
static_assert(STACK_LDS >= 1 && STACK_LDS <= 63 && STACK_LDS * BLOCK * 12 <= 160 * 1024, "MIPT_STACK_LDS: 1..63 entries, and the block's stack must fit the CU's LDS");

// ---- float planes
enum : int {
    P_FILMX = 0, P_FILMY,
    P_TENC0, P_TENC1, P_TENC2, P_TENC3,   // the ray's tMax when the k-th postponed quadric was met (closest-hit traversals)
    P_COUNT
};
// ---- float4 record planes: values that are read and written together travel in one 16-B access
enum : int {
    R_RAY0 = 0,   // path ray: o.xyz, tMax
    R_RAY1,       //           d.xyz, etaScale
    R_SH0, R_SH1, // shadow ray (tMax = 1 - ShadowEpsilon): o.xyz d.x | d.y d.z - -
    R_MI0, R_MI1, // MIS ray: same packing
    R_HIT,        // t, b0, b1, b2 of the last traversal (path ray, or MIS ray)
    R_COUNT
};
// The two quads of a ray are neighbours in memory ([ray][slot][2]: 32 B per slot), so the pair a shading lane stores --
// and a traversal lane loads -- falls into ONE 64-B sector instead of two planes' sectors (the memory side moves whole
// sectors: a scattered 16-B store cost ~45 B there, PMC of round 1).
// ---- spectral planes. A 31-bin spectrum of a slot is stored as NQ = 8 float4 "quad planes": bins
// 4c..4c+3 of slot i at q[(set + c) * pool + i] (bin 31 is padding, kept 0 in everything that is summed or
// tested). One 16-B access per lane moves four bins, so a spectral pass issues a quarter of the memory
// instructions of one-float planes and -- the lanes of a shading wave hold scattered slots -- touches fewer
// cache lines per bin.
constexpr int NQ = 8;
enum : int {
    Q_L = 0,                // radiance gathered by the path
    Q_BETA = NQ,            // throughput
    Q_LB = 2 * NQ,          // the second line of the path's radiance (F_L_IN_B, F_CAND)
    Q_LMIS = 3 * NQ,        // pending BSDF-sample (MIS) contribution
    Q_COUNT = 4 * NQ,
    Q_LCA = Q_COUNT         // Integrator "spectralpath" only: the sample's stitched bands
};
DEV float Get4(const float4 &v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
DEV void Set4(float4 &v, int k, float x) { if (k == 0) v.x = x; else if (k == 1) v.y = x; else if (k == 2) v.z = x; else v.w = x; }
// ---- int planes
enum : int { I_HITPRIM = 0, I_PIXEL, I_SAMPLE, I_IDXLO, I_IDXHI, I_DIM /* pixel samplers only: see StateWord */, I_FLAGS, I_MISLIGHT,
             I_NPEND, I_PEND0, I_PEND1, I_PEND2, I_PEND3,  // quadrics postponed by the traversal kernel
             I_BAND,                                       // spectralpath: band (path number) of the camera sample
             I_HITINST,                                    // instance the hit primitive was reached through, -1 = none (scenes with instances)
             I_COUNT };
constexpr int MAX_PEND = 4;
constexpr int PEND_OVERFLOW = 0x100;  // more quadrics met than MAX_PEND: the resolve kernel re-traverses
// ---- slot flags
enum : int {
    F_ALIVE = 1, F_FINISHED = 2, F_SPECULAR = 4, F_SHADOW = 8, F_MIS = 16, F_NEE = 32, F_A_ADDED = 64,
    // The path's radiance L has two lines, Q_L and Q_LB, and this bit says which one holds it. A light sample's contribution
    // used to wait in a line of its own until its shadow ray was resolved, and k_resolve_shadow then read it, read L and
    // wrote L: 384 B per unoccluded ray in a kernel that does nothing else. Now k_shade writes the CANDIDATE L + contribution
    // into the line that does not hold L (F_CAND; it has the throughput and the contribution in hand, and L costs it one
    // more read -- none while L is still empty), and an unoccluded ray flips this bit: the resolve moves no spectrum at all.
    F_L_IN_B = 128,
    // A new path has L = 0 and beta = 1. Writing those 16 quads into freshly (sparsely) refilled slots cost as
    // much as the rest of k_generate, so they stay implicit until something else is written there:
    F_L_ZERO = 256,     // Q_L holds no value yet; it reads as 0 (0 + x == x)
    F_BETA_ONE = 512,   // Q_BETA holds no value yet; it reads as 1 (1 * x == x)
    F_DIFF = 1024,      // the path ray is still the camera ray: it has ray differentials (RayDifferential::hasDifferentials)
    // F_CAND: the line that does not hold L holds L + the pending light sample's contribution (see F_L_IN_B); the shadow ray
    // decides. F_NEE_NZ: that contribution has a non-zero bin.
    F_CAND = 2048, F_NEE_NZ = 4096,
    // The BSDF-sampled (MIS) ray is traced, but it cannot reach the sampled area light (it misses the dilated bounds of the
    // light's shape: Sphere::Pdf gives every direction the cone's pdf, sphere.cpp:294-310, so the estimate goes on for rays
    // that point away from the sphere), so its contribution was neither formed nor stored in Q_LMIS.
    F_MIS_DARK = 8192
};
// The I_FLAGS word carries the path's bounce count and sampler dimension beside the flags: bits 0-13 the flags above, 14-21
// `bounces` (mi_pt_create bounds max_depth by 255), 22-31 the next sampler dimension (the Halton / Sobol' tables end at 1000 /
// 1024 dimensions; the random sampler only counts). Three 4-byte planes used to hold them: a shading lane read and wrote
// all three at scattered slots (~45 B each way at the memory side, PMC of round 2), k_generate stored all three per new
// path. One word: every kernel that reads the flags has the other two for nothing. (A pixel sampler's two table counters
// need 32 bits: they stay in the I_DIM plane.)
constexpr int FLAG_BITS = 14, FLAG_MASK = (1 << FLAG_BITS) - 1, BOUNCE_SHIFT = 14, DIM_SHIFT = 22;
static_assert(F_MIS_DARK < (1 << FLAG_BITS), "slot flags outgrew their field");
DEV int LPlane(int flags) { return (flags & F_L_IN_B) ? Q_LB : Q_L; }        // the line that holds the path's L
DEV int LOtherPlane(int flags) { return (flags & F_L_IN_B) ? Q_L : Q_LB; }   // ... and the one for the candidate
DEV int StateWord(int flags, int bounces, int dim) { return (flags & FLAG_MASK) | ((bounces & 0xff) << BOUNCE_SHIFT) | (int)((unsigned)(dim & 0x3ff) << DIM_SHIFT); }
DEV int StateBounces(int word) { return (word >> BOUNCE_SHIFT) & 0xff; }
DEV int StateDim(int word) { return (int)((unsigned)word >> DIM_SHIFT); }
// Conservative: false only if the ray o + t d, t >= 0, stays outside the box; [*lo, *hi] contains the parameters at which it is inside.
DEV bool RaySpanInBox(const V3 &o, const V3 &d, const float4 &bmin, const float4 &bmax, float *spanLo, float *spanHi) {
    float t0 = 0.f, t1 = kInfinity;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, lo[3] = {bmin.x, bmin.y, bmin.z}, hi[3] = {bmax.x, bmax.y, bmax.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (dd[a] == 0.f) { if (oo[a] < lo[a] || oo[a] > hi[a]) return false; continue; }
        const float inv = 1.f / dd[a];
        float ta = (lo[a] - oo[a]) * inv, tb = (hi[a] - oo[a]) * inv;
        if (ta > tb) { const float tt = ta; ta = tb; tb = tt; }
        tb *= 1.0001f;   // (rounding of the products: widen the exit)
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    *spanLo = t0; *spanHi = t1;
    return !(t0 > t1);
}
constexpr unsigned MIS_EXCL_BITS = 27, MIS_EXCL_NONE = (1u << MIS_EXCL_BITS) - 1u;
// The two words k_trav<3> reads beside a MIS ray (R_MI1.z, .w): the end tHi of the span in which the sampled light's shape can
// be hit, and that shape's primitive | c << 27 with tLo = tHi (1 - 2^-c) at or below the span's start (c = 0: tLo = 0).
DEV void MisSpanWords(float lo, float hi, unsigned prim, float *tHi, float *word) {
    const float w = 1.f - lo / hi;          // tLo = hi (1 - 2^-c) <= lo  <=>  2^-c >= w
    int c = 126 - (int)((__float_as_uint(w) >> 23) & 0xffu);
    if (!(w > 0.f)) c = 20;
    c = (lo > 0.f && hi < kInfinity) ? max(0, min(c, 20)) : 0;   // (1 - 2^-c is exact in a float up to c = 24)
    *tHi = hi;
    *word = __uint_as_float(prim | ((unsigned)c << MIS_EXCL_BITS));
}

// Shading classes: materials with the same lobe-type list share a class (ids in order of
// first appearance, the 15th and later share class 14); class 15 holds the vertices without a
// BSDF (escaped rays, interface primitives). A wave of k_shade works on one class only.
constexpr int MAX_CLASSES = 16;
constexpr int MISS_CLASS = 15;

struct Pool {
    float *f;
    float4 *q;   // spectral quad planes
    float4 *r;   // record planes
    int *i;
    uint32_t *shadowQ, *misQ;  // compacted slot indices of this iteration's shadow / MIS rays. shadowQ[n + k]: the any-hit
                               // traversal's answer for entry k (bit 31: occluded; below it the count of postponed quadrics |
                               // PEND_OVERFLOW) -- in queue order, so k_resolve_shadow reads it as whole lines where the
                               // hit words of the slot planes cost it a sector each. misQ[n + 2k], [n + 2k + 1]: hit
                               // primitive and postponed quadrics of MIS ray k, likewise (its t and barycentrics, which
                               // k_resolve_mis rarely needs, stay in the slot's R_HIT)
    uint32_t *extQ;            // this iteration's path rays: new camera rays from the front (coherent: consecutive
                               // samples of a pixel), continuing paths from the back
    uint32_t *shadeQ;          // slots to shade: MAX_CLASSES queues of n entries, one per shading class
    uint32_t *ovfQ;            // rays whose list of postponed quadrics overflowed: 3 queues of n entries (extend / shadow / MIS)
    uint32_t n;
    DEV float &F(int plane, uint32_t slot) const { return f[(size_t)plane * n + slot]; }
    // a slot's 8 quads of one spectrum are one 128-B line: [spectrum][slot][quad] ([slot][spectrum][quad], the
    // spectra of a slot in one page, measured the same)
    DEV float4 &Q(int plane, uint32_t slot) const { return q[(((size_t)(plane >> 3) * n + slot) << 3) + (plane & 7)]; }
    DEV float4 &R(int plane, uint32_t slot) const {
        return plane < R_HIT ? r[(((size_t)(plane >> 1) * n + slot) << 1) + (plane & 1)] : r[(size_t)plane * n + slot];
    }
    DEV int &I(int plane, uint32_t slot) const { return i[(size_t)plane * n + slot]; }
};

// Statistics are striped: STAT_STRIPES copies, each on its own 128-B line, picked by block
// index, so the per-wave atomics of a launch do not all serialise on one L2 line; the host
// adds the stripes up. The queue cursors cannot be striped (they hand out consecutive
// positions); they are bumped once per block instead (BlockReserve) and sit on lines of
// their own.
constexpr int STAT_STRIPES = 64;
struct alignas(128) DevStats {
    unsigned long long cameraRays, regularRays, shadowRays, totalPaths, zeroRadiancePaths, pathLengthSum, nodesVisited,
        triTests, badSamples, extendNodes, extendTris, extendRays;
};
struct alignas(128) DevCursor {
    unsigned int v;
};
struct DevCounters {
    DevStats stats[STAT_STRIPES];
    alignas(128) unsigned long long nextWork;   // global work counter
    // cleared together every iteration:
    DevCursor alive;                    // slots alive after generate
    DevCursor shadowCount, misCount;    // entries in Pool::shadowQ / misQ
    DevCursor primCount, contCount;     // entries at the front / back of Pool::extQ
    DevCursor shadeCount[MAX_CLASSES];  // entries in shading queue c
    DevCursor travNext[3];              // work cursors of the persistent traversal kernels (extend/shadow/mis)
    DevCursor ovfCount[3];              // entries in Pool::ovfQ (extend/shadow/mis)
#ifdef MIPT_EXP_STAMPS
    unsigned long long phase[24];       // diagnostic build: wave-cycles of k_shade between its stamps (s_memtime), summed over waves
    unsigned long long phaseWaves;
#endif
};
#ifdef MIPT_EXP_STAMPS
// In-kernel stamps (MI355X_MICROARCH.md, "in-kernel stamps"): one wave-uniform s_memtime per phase boundary, the difference
// to the previous stamp added to phase[k]. Diagnostic build only (tools/shade_experiments.sh stamps).
#define STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        if (stampOn && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&ctr->phase[k], t_ - stampLast); stampLast = t_; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
constexpr size_t ITER_CLEAR_BYTES = sizeof(DevCursor) * (5 + MAX_CLASSES + 3 + 3);
DEV DevStats &Stats(DevCounters *ctr) { return ctr->stats[blockIdx.x & (STAT_STRIPES - 1)]; }

struct WorkDesc {
    unsigned long long totalWork;
    int nTilesX, nTilesY, nTilesShard, shardIndex, shardCount;
    long long spp, sampleBegin;
    int run;   // consecutive samples of a pixel handed out together (a power of two dividing spp, at most MIPT_WORK_RUN = 16)
};

DEV unsigned long long WaveSum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
DEV void CountAdd(unsigned long long *ctr, unsigned long long v) {
    v = WaveSum(v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(ctr, v);
}

// Wave-level compaction: every lane of the wave calls this (convergent); lanes with
// pred get consecutive queue positions from one atomic per wave (ballot + popcount).
DEV void QueueAppend(unsigned *counter, uint32_t *queue, bool pred, uint32_t value) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (pred) queue[base + __popcll(mask & ((1ull << lane) - 1))] = value;
}

// Block-level compaction: every thread of the block calls this (it synchronises); threads
// with pred get consecutive positions from ONE atomicAdd per block. `scratch` is 5 words of
// LDS that no other call of the same kernel uses.
DEV unsigned BlockReserve(unsigned *counter, bool pred, unsigned *scratch) {
    const unsigned long long mask = __ballot(pred);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) scratch[wave] = (unsigned)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tot = scratch[0] + scratch[1] + scratch[2] + scratch[3];
        scratch[4] = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    unsigned base = scratch[4];
    for (int w = 0; w < wave; ++w) base += scratch[w];
    return base + (unsigned)__popcll(mask & ((1ull << lane) - 1));
}

// Two reservations behind ONE pair of barriers, their atomics issued by two different waves (k_shade ends in an append to the
// shadow queue and one to the MIS queue; one after the other they cost two atomic round trips and four barriers).
// `scratch`: 10 words of LDS.
DEV void BlockReserve2(unsigned *counterA, bool predA, unsigned *counterB, bool predB, unsigned *scratch, unsigned *posA, unsigned *posB) {
    const unsigned long long mA = __ballot(predA), mB = __ballot(predB);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { scratch[wave] = (unsigned)__popcll(mA); scratch[5 + wave] = (unsigned)__popcll(mB); }
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == 64) {
        unsigned *sc = scratch + (threadIdx.x ? 5 : 0);
        const unsigned tot = sc[0] + sc[1] + sc[2] + sc[3];
        sc[4] = tot ? atomicAdd(threadIdx.x ? counterB : counterA, tot) : 0;
    }
    __syncthreads();
    unsigned a = scratch[4], b = scratch[9];
    for (int w = 0; w < wave; ++w) { a += scratch[w]; b += scratch[5 + w]; }
    const unsigned long long lt = (1ull << lane) - 1ull;
    *posA = a + (unsigned)__popcll(mA & lt);
    *posB = b + (unsigned)__popcll(mB & lt);
}

// ------------------------------------------------------------------ traversal
struct Hit {
    int prim;
    float t, b0, b1, b2;
    int inst = -1;   // the object instance the hit primitive was reached through (mi_instance index), -1: a world primitive
};

// BVHAccel::Intersect / IntersectP (bvh.cpp:662-738) with Bounds3::IntersectP
// (geometry.h:1420-1447) over the "wide" node array: an interior node carries the boxes
// of both children, so one 64-B fetch decides both child visits and the ray's chain of
// dependent loads is halved. The visit semantics are the reference's:
//   * near child (by dirIsNeg[axis]) first; the far child is pushed only if its box is
//     hit, together with its entry distance tMin;
//   * when popped, the far child is re-validated with `tMin < tMax` -- the only part of
//     the reference's box test that depends on the (shrinking) ray tMax -- so exactly the
//     nodes the reference would enter are entered, and leaf primitives are tested in the
//     same order against the same tMax: equal-t ties resolve identically.
// SIMT schedule: "while-while" (walk interior nodes until the lane holds a leaf, then the
// wave tests leaves together). Quadric primitives are postponed: recorded in encounter
// order with the ray's tMax of that moment and replayed after the triangles in the
// reference's order (ResolveQuadrics), so the interval-arithmetic sphere code runs at full
// lane utilisation.
// (TMIN = false: the any-hit kernel keeps no entry distances -- a shadow ray's tMax never shrinks, so an entry pushed in
// front of it stays in front of it -- which takes the stack from 36 to 24 KB per block: five blocks per CU instead of four)
template <bool TMIN>
struct TravLdsT {
    int node[STACK_LDS][BLOCK];
    int meta[STACK_LDS][BLOCK];
    float tmin[TMIN ? STACK_LDS : 1][BLOCK];
};
// The block's traversal stack lives in one LDS object reached by name (no generic pointers).
template <bool TMIN>
DEV TravLdsT<TMIN> &TravStack() {
    __shared__ TravLdsT<TMIN> stack;
    return stack;
}
struct TravSpill {
    int node[STACK_SPILL];
    int meta[STACK_SPILL];
    float tmin[STACK_SPILL];
};
struct RayCtx {  // per-lane ray constants
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
    bool n0, n1, n2;
    bool slow;   // a component of the direction is zero (an infinite reciprocal): only then can a slab product be NaN (BoxTestFast)
    DEV int Octant() const { return (n0 ? 1 : 0) | (n1 ? 2 : 0) | (n2 ? 4 : 0); }
};
DEV void InitRayCtx(RayCtx &r, float ox, float oy, float oz, float dx, float dy, float dz) {
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = 1.f / dx; r.iy = 1.f / dy; r.iz = 1.f / dz;
    r.n0 = r.ix < 0; r.n1 = r.iy < 0; r.n2 = r.iz < 0;
    r.slow = !(absf(r.ix) < kInfinity && absf(r.iy) < kInfinity && absf(r.iz) < kInfinity);
}
// Bounds3::IntersectP(ray, invDir, dirIsNeg); also returns the final tMin.
DEV bool BoxTest(const RayCtx &r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float tMaxRay, float *tMinOut) {
    const float k = 1 + 2 * gammaf(3);
    float tMin = ((r.n0 ? mxx : mnx) - r.ox) * r.ix;
    float tMx = ((r.n0 ? mnx : mxx) - r.ox) * r.ix;
    float tyMin = ((r.n1 ? mxy : mny) - r.oy) * r.iy;
    float tyMax = ((r.n1 ? mny : mxy) - r.oy) * r.iy;
    tMx *= k;
    tyMax *= k;
    bool hit = !(tMin > tyMax || tyMin > tMx);
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMx) tMx = tyMax;
    float tzMin = ((r.n2 ? mxz : mnz) - r.oz) * r.iz;
    float tzMax = ((r.n2 ? mnz : mxz) - r.oz) * r.iz;
    tzMax *= k;
    hit = hit && !(tMin > tzMax || tzMin > tMx);
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMx) tMx = tzMax;
    *tMinOut = tMin;
    return hit && (tMin < tMaxRay) && (tMx > 0);
}

// The same test for a ray whose reciprocal direction is finite (RayCtx::slow == false). Then no slab product is NaN (a
// finite difference times a finite factor), and without NaNs the reference's selects and compare-and-assign steps are
// minima and maxima: the near plane's product is the smaller of the two (rounding is monotone), `if (tyMin > tMin) tMin =
// tyMin` is a maximum, and the two early-outs together reject exactly the rays whose largest entry exceeds their smallest
// scaled exit (the x-y test failing implies the x-y-z test failing). Same products, same comparisons, same tMin: 29
// instructions a box instead of 42 (six selects and four compare-and-select pairs become three v_min, three v_max, one
// v_max3 and one v_min3). With a zero direction component (0 * inf = NaN when the origin lies in a slab plane) the NaN
// takes the reference's path through the comparisons, and the wave takes BoxTest.
DEV bool BoxTestFast(const RayCtx &r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float tMaxRay, float *tMinOut) {
    const float k = 1 + 2 * gammaf(3);
    const float ax = (mnx - r.ox) * r.ix, bx = (mxx - r.ox) * r.ix;
    const float ay = (mny - r.oy) * r.iy, by = (mxy - r.oy) * r.iy;
    const float az = (mnz - r.oz) * r.iz, bz = (mxz - r.oz) * r.iz;
    const float tMin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tMx = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx) * k, __builtin_fmaxf(ay, by) * k), __builtin_fmaxf(az, bz) * k);
    *tMinOut = tMin;
    return (tMin <= tMx) && (tMin < tMaxRay) && (tMx > 0);
}

// One traversal state machine step set, shared by the persistent kernel and the plain
// per-ray routine. cur >= 0: interior node to open; -2: take the next stack entry; -1: done.
struct TravState {
    int cur, sp;
};
template <bool TMIN = true>
DEV void StackPush(TravSpill &sp, int lane, int &n, int node, int meta, float tmin) {
    TravLdsT<TMIN> &lds = TravStack<TMIN>();
    if (n >= STACK_LDS + STACK_SPILL) return;   // (cannot happen: mi_pt_create bounds the depth of the tree it uploads)
    if (n < STACK_LDS) { lds.node[n][lane] = node; lds.meta[n][lane] = meta; if (TMIN) lds.tmin[n][lane] = tmin; }
    else {
        // keep the LDS and the scratch path apart: merged into one store through a selected
        // generic pointer, hipcc 7.2 emits an illegal address-space test for gfx950
        sp.node[n - STACK_LDS] = node; sp.meta[n - STACK_LDS] = meta; sp.tmin[n - STACK_LDS] = tmin;
        asm volatile("" ::: "memory");
    }
    ++n;
}
template <bool TMIN = true>
DEV void StackPop(TravSpill &sp, int lane, int &n, int *node, int *meta, float *tmin) {
    TravLdsT<TMIN> &lds = TravStack<TMIN>();
    --n;
    if (n < STACK_LDS) { *node = lds.node[n][lane]; *meta = lds.meta[n][lane]; *tmin = TMIN ? lds.tmin[n][lane] : 0.f; }
    else {
        int nd = sp.node[n - STACK_LDS], mt = sp.meta[n - STACK_LDS];
        float tm = sp.tmin[n - STACK_LDS];
        asm volatile("" : "+v"(nd), "+v"(mt), "+v"(tm));  // pins the scratch loads in this branch
        *node = nd; *meta = mt; *tmin = tm;
    }
}
// Open interior node `cur`: test its children's boxes, take the first one hit in the reference's visiting order, push the
// others (with their entry distances) so that they pop in that order. Returns whether a child was taken.
//
// W = 2: a record per BVH2 interior node (both children's boxes). W = 4: a record per two-level subtree of the BVH2 -- the
// boxes of up to four grandchildren (a child that is a leaf stands for itself) and, for each of the 8 sign combinations of
// the ray direction, the order in which the reference's traversal would reach them (near child first at the subtree's
// root by its split axis, then near grandchild first inside each child by the child's axis; bvh.cpp:686-692). The child's
// own box is not tested: a ray that hits a grandchild's box hits the child's box (it contains it, and the slab arithmetic
// is monotone), and the entry-distance re-validation at pop time is the only tMax-dependent part of the test. So the lane
// enters the same leaves in the same order against the same tMax as the BVH2 traversal -- closest hits, equal-t ties and
// the count of primitive tests are unchanged -- with half the dependent node fetches per ray.
template <int W, bool TMIN = true, bool FAST = false>
DEV bool OpenNode(const float4 *__restrict__ wnodes, int cur, const RayCtx &r, float tMax, TravSpill &spill, int lane, int &sp,
                  int *tkChild, int *tkMeta, unsigned &nodeCount) {
    if constexpr (W == 2) {
        const float4 a = wnodes[4 * cur], b = wnodes[4 * cur + 1], c = wnodes[4 * cur + 2];
        const float4 dd = wnodes[4 * cur + 3];
        const int childL = __float_as_int(dd.x), childR = __float_as_int(dd.y);
        const int metaL = __float_as_int(dd.z), metaR = __float_as_int(dd.w);
        const bool haveR = (metaR & 0xffff) != 0xffff;
        float tL, tR = 0;
        const bool hitL = BoxTest(r, a.x, a.y, a.z, a.w, b.x, b.y, tMax, &tL);
        const bool hitR = haveR && BoxTest(r, b.z, b.w, c.x, c.y, c.z, c.w, tMax, &tR);
        nodeCount += haveR ? 2 : 0;
        const int axis = (metaL >> 16) & 0xff;
        const bool negAxis = (axis == 0) ? r.n0 : ((axis == 1) ? r.n1 : r.n2);
        // near child first (bvh.cpp:686-692): left unless the ray runs against the split axis
        const bool hitF = negAxis ? hitR : hitL, hitS = negAxis ? hitL : hitR;
        const int chF = negAxis ? childR : childL, chS = negAxis ? childL : childR;
        const int mtF = (negAxis ? metaR : metaL) & 0xffff, mtS = (negAxis ? metaL : metaR) & 0xffff;
        const float tS = negAxis ? tL : tR;
        if (hitF) {
            *tkChild = chF; *tkMeta = mtF;
            if (hitS) StackPush<TMIN>(spill, lane, sp, chS, mtS, tS);
            return true;
        }
        if (hitS) { *tkChild = chS; *tkMeta = mtS; return true; }
        return false;
    } else {
        const float4 *__restrict__ n = wnodes + 8 * (size_t)cur;
        const float4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3], q4 = n[4], q5 = n[5], q6 = n[6], q7 = n[7];
        const unsigned c01 = __float_as_uint(q7.x), c23 = __float_as_uint(q7.y);
        const int cnt0 = (int)(c01 & 0xffffu), cnt1 = (int)(c01 >> 16), cnt2 = (int)(c23 & 0xffffu), cnt3 = (int)(c23 >> 16);
        const bool have1 = cnt1 != 0xffff, have2 = cnt2 != 0xffff, have3 = cnt3 != 0xffff;
        float t0, t1 = 0, t2 = 0, t3 = 0;
        bool h0, h1, h2, h3;
        if constexpr (FAST) {   // (every walking lane's reciprocal direction is finite: the caller asked the wave)
            h0 = BoxTestFast(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tMax, &t0);
            h1 = BoxTestFast(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tMax, &t1) && have1;
            h2 = BoxTestFast(r, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, tMax, &t2) && have2;
            h3 = BoxTestFast(r, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, tMax, &t3) && have3;
        } else {
            h0 = BoxTest(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tMax, &t0);
            h1 = have1 && BoxTest(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tMax, &t1);
            h2 = have2 && BoxTest(r, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, tMax, &t2);
            h3 = have3 && BoxTest(r, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, tMax, &t3);
        }
        nodeCount += 1u + (have1 ? 1u : 0u) + (have2 ? 1u : 0u) + (have3 ? 1u : 0u);
        const unsigned hm = (h0 ? 1u : 0u) | (h1 ? 2u : 0u) | (h2 ? 4u : 0u) | (h3 ? 8u : 0u);
        if (hm == 0u) return false;
        // (an any-hit ray -- TMIN = false -- may take the children in any order: whether something occludes it does not
        // depend on it, so it takes them as they lie in the record and skips the octant's visiting order: shadow launches -6 %)
        unsigned perm = 0xE4u;   // slot visited k-th at bits 2k..2k+1
        if constexpr (TMIN) {
            const int oct = r.Octant();
            const unsigned ordWord = (oct & 4) ? __float_as_uint(q7.w) : __float_as_uint(q7.z);
            perm = (ordWord >> (8 * (oct & 3))) & 0xffu;
        }
        const int l0 = __float_as_int(q6.x), l1 = __float_as_int(q6.y), l2 = __float_as_int(q6.z), l3 = __float_as_int(q6.w);
        // from the last visited to the first: a slot that is hit displaces the candidate found so far onto the stack, so the
        // stack receives the later ones first and the first one in order stays in hand
        int cand = -1;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const int sl = (int)((perm >> (2 * k)) & 3u);
            if ((hm >> sl) & 1u) {
                if (cand >= 0) {
                    const int lk = cand == 0 ? l0 : (cand == 1 ? l1 : (cand == 2 ? l2 : l3));
                    const int ct = cand == 0 ? cnt0 : (cand == 1 ? cnt1 : (cand == 2 ? cnt2 : cnt3));
                    const float tt = cand == 0 ? t0 : (cand == 1 ? t1 : (cand == 2 ? t2 : t3));
                    StackPush<TMIN>(spill, lane, sp, lk, ct, tt);
                }
                cand = sl;
            }
        }
        *tkChild = cand == 0 ? l0 : (cand == 1 ? l1 : (cand == 2 ? l2 : l3));
        *tkMeta = cand == 0 ? cnt0 : (cand == 1 ? cnt1 : (cand == 2 ? cnt2 : cnt3));
        return true;
    }
}

// A leaf's primitive count in the traversal records carries LEAF_SIMPLE when every primitive of the leaf is a triangle (no
// quadric to postpone, no instance to enter): those leaves go through the cooperative test of k_trav.
constexpr int LEAF_SIMPLE = 0x4000, LEAF_COUNT_MASK = 0x3fff;

// Advance until the lane holds a leaf (returns true with leafOffset/leafCount) or the
// traversal is finished (returns false, st.cur == -1).
template <int W>
DEV bool NextLeaf(const float4 *__restrict__ wnodes, const RayCtx &r, float tMax, TravState &st, TravSpill &spill,
                  int lane, int *leafOffset, int *leafCount, unsigned &nodeCount, int floor = 0) {
    while (st.cur != -1) {
        int tkChild = 0, tkMeta = 0;
        bool got = false;
        if (st.cur >= 0) got = OpenNode<W>(wnodes, st.cur, r, tMax, spill, lane, st.sp, &tkChild, &tkMeta, nodeCount);
        while (!got && st.sp > floor) {  // a popped node is entered only if still in front of tMax
            float t;
            StackPop(spill, lane, st.sp, &tkChild, &tkMeta, &t);
            got = t < tMax;
        }
        if (!got) { st.cur = -1; return false; }
        if (tkMeta > 0) { *leafOffset = tkChild; *leafCount = tkMeta & LEAF_COUNT_MASK; st.cur = -2; return true; }
        st.cur = tkChild;
    }
    return false;
}
DEV void StartTraversal(const DScene &s, const RayCtx &r, float tMax, TravState &st, unsigned &nodeCount) {
    st.sp = 0;
    st.cur = -1;
    if (s.nNodes == 0) return;
    float t;
    ++nodeCount;
    if (BoxTest(r, s.wbMin[0], s.wbMin[1], s.wbMin[2], s.wbMax[0], s.wbMax[1], s.wbMax[2], tMax, &t)) st.cur = 0;
}

constexpr int MAX_PENDING_SPHERES = 3;

// Plain per-ray traversal with inline quadric tests (mi_pt_trace and the overflow path
// of ResolveQuadrics). TOP: the world's tree, whose leaves may hold object instances -- a TransformedPrimitive
// (primitive.cpp:78-99) takes the ray to the instance's space and traverses the object's tree with it, above the entries
// the world's traversal has on the stack (`floor`); a hit inside hands its tMax back to the world ray.
template <bool ANY, int W, bool TOP>
DEV bool TraverseTree(const DScene &s, int rootRecord, bool rootHit, const V3 &ro, const V3 &rd, float &tMax, int floor, int inst,
                      Hit *hit, unsigned &nodeCount, unsigned &triCount) {
    const int lane = threadIdx.x;
    RayCtx r;
    InitRayCtx(r, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
    TravSpill spill;
    TravState st;
    st.sp = floor;
    st.cur = rootHit ? rootRecord : -1;
    bool found = false;
    const float4 *__restrict__ primTri = s.primTri;
    int leafOffset = 0, leafCount = 0;
    while (NextLeaf<W>(s.wnodes, r, tMax, st, spill, lane, &leafOffset, &leafCount, nodeCount, floor)) {
        for (int i = 0; i < leafCount; ++i) {
            const int prim = leafOffset + i;
            const float4 v0 = primTri[3 * prim];
            const unsigned pf = __float_as_uint(v0.w);
            if constexpr (TOP && W == 4) {
                if (pf & PRIM_FLAG_INSTANCE) {
                    const int k = __float_as_int(primTri[3 * prim + 1].w);
                    const Ray ir = XfRay(s.instances[k].w2i, Ray(ro, rd, tMax));   // Inverse(InstanceToWorld)(r)
                    RayCtx rc;
                    InitRayCtx(rc, ir.o.x, ir.o.y, ir.o.z, ir.d.x, ir.d.y, ir.d.z);
                    const float4 bMin = s.instRootBounds[2 * k], bMax = s.instRootBounds[2 * k + 1];
                    float tBox, tIn = ir.tMax;
                    ++nodeCount;
                    const bool boxHit = BoxTest(rc, bMin.x, bMin.y, bMin.z, bMax.x, bMax.y, bMax.z, tIn, &tBox);
                    if (TraverseTree<ANY, W, false>(s, s.instWideRoot[k], boxHit, ir.o, ir.d, tIn, st.sp, k, hit, nodeCount, triCount)) {
                        if (ANY) return true;
                        tMax = tIn;   // r.tMax = ray.tMax
                        found = true;
                    }
                    continue;
                }
            }
            if (pf & PRIM_FLAG_SPHERE) {   // tested where it is met, against the tMax of that moment (the reference's order)
                const int sph = __float_as_int(primTri[3 * prim + 1].w);
                float t;
                if (SphereHitT(s.spheres[sph], ro, rd, tMax, &t)) {
                    if (ANY) return true;
                    tMax = t;
                    hit->prim = prim; hit->t = t; hit->b0 = hit->b1 = hit->b2 = 0; hit->inst = inst;
                    found = true;
                }
                continue;
            }
            const float4 v1 = primTri[3 * prim + 1], v2 = primTri[3 * prim + 2];
            ++triCount;
            TriHit th;
            if (TriTest(V3(v0.x, v0.y, v0.z), V3(v1.x, v1.y, v1.z), V3(v2.x, v2.y, v2.z), ro, rd, tMax, &th)) {
                if ((pf & PRIM_FLAG_ALPHA) &&
                    ((pf & PRIM_FLAG_DEGENERATE) || !AlphaPass(s, __float_as_int(v1.w), th.b0, th.b1, th.b2, ANY))) continue;
                if (ANY) return true;
                if (!(pf & PRIM_FLAG_DEGENERATE)) {
                    tMax = th.t;
                    hit->prim = prim; hit->t = th.t; hit->b0 = th.b0; hit->b1 = th.b1; hit->b2 = th.b2; hit->inst = inst;
                    found = true;
                }
            }
        }
    }
    return found;
}
// INST: compiled for scenes with object instances (the nested traversal costs the calling kernel ~40 VGPRs)
template <bool ANY, int W, bool INST>
DEV bool TraverseW(const DScene &s, const V3 &ro, const V3 &rd, float tMax, Hit *hit, unsigned &nodeCount, unsigned &triCount) {
    RayCtx r;
    InitRayCtx(r, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
    TravState st;
    StartTraversal(s, r, tMax, st, nodeCount);
    return TraverseTree<ANY, W, INST>(s, 0, st.cur >= 0, ro, rd, tMax, 0, -1, hit, nodeCount, triCount);
}

template <bool ANY, bool INST>
DEV bool Traverse(const DScene &s, const V3 &ro, const V3 &rd, float tMax, Hit *hit, unsigned &nodeCount, unsigned &triCount) {
    if (s.bvhWidth == 4) return TraverseW<ANY, 4, INST>(s, ro, rd, tMax, hit, nodeCount, triCount);
    return TraverseW<ANY, 2, false>(s, ro, rd, tMax, hit, nodeCount, triCount);
}

DEV void HitInteraction(const DScene &s, int prim, const V3 &ro, const V3 &rd, float b0, float b1, float b2, SurfaceInteraction *si);

// ------------------------------------------------------------------ persistent traversal
// One launch traverses every ray of a class (MODE 0: path rays of the alive slots, closest
// hit; 1: NEE shadow rays of shadowQ, any hit; 2: MIS rays of misQ, closest hit; 3: the same rays as visibility queries,
// see TRAV_IS_ANY below). A fixed
// grid of waves pulls rays from a device-wide cursor: lanes whose ray finished fetch a new
// one as soon as fewer than REFILL_BELOW lanes of the wave are still traversing (dynamic
// fetch), so a long ray no longer idles the other 63 lanes. Quadrics are only recorded
// (I_PEND*); k_resolve_* tests them afterwards at full lane utilisation.
#ifndef MIPT_TRAV_BLOCKS_PER_CU
#define MIPT_TRAV_BLOCKS_PER_CU 4
#endif
#ifndef MIPT_REFILL_BELOW
#define MIPT_REFILL_BELOW 44   // (same-box A/B with the two-level records: 44 against 32, killeroo +1 %, the 10M-triangle scene +2 %)
#endif
constexpr int TRAV_BLOCKS_PER_CU = MIPT_TRAV_BLOCKS_PER_CU;
constexpr int REFILL_BELOW = MIPT_REFILL_BELOW;
#ifndef MIPT_TRI_BATCH
#define MIPT_TRI_BATCH 16
#endif
constexpr int TRI_BATCH = MIPT_TRI_BATCH;
#ifndef MIPT_COOP_KMAX
#define MIPT_COOP_KMAX 4
#endif
static_assert(MIPT_TRAV_BLOCKS_PER_CU >= 1 && MIPT_TRAV_BLOCKS_PER_CU <= 8 && MIPT_REFILL_BELOW >= 1 && MIPT_REFILL_BELOW <= 64 && MIPT_TRI_BATCH >= 1 &&
              MIPT_TRI_BATCH <= 64 && MIPT_COOP_KMAX >= 1 && MIPT_COOP_KMAX <= 8, "traversal tuning macros out of range");
constexpr int COOP_KMAX = MIPT_COOP_KMAX;   // (triangle, ray) pairs a waiting lane hands to the wave per pass (pbrt's default leaf size)
#ifndef MIPT_TRAV_CHUNK
#define MIPT_TRAV_CHUNK 128
#endif
static_assert(MIPT_TRAV_CHUNK >= 64 && MIPT_TRAV_CHUNK <= (1 << 20), "MIPT_TRAV_CHUNK: at least a wave's worth of entries per cursor atomic");
constexpr int TRAV_CHUNK = MIPT_TRAV_CHUNK;  // work-list entries a wave reserves per cursor atomic (twice that for launches of
                                             // 8M rays and more: the cursor word takes ~88 adds/us, and 30M rays in chunks of
                                             // 128 are 234k adds -- same-box A/B: killeroo +3.7 % with 256, the 10M-triangle
                                             // scene's 6M-ray launches -1 %, so the size follows the launch)

#ifndef MIPT_TRAV_WAVES_PER_EU
#define MIPT_TRAV_WAVES_PER_EU 4
#endif
#ifndef MIPT_TRAV_WAVES_PER_EU_ANY
#define MIPT_TRAV_WAVES_PER_EU_ANY 5   // the any-hit kernel (without the alpha-mask code): 96 VGPRs and a 24-KB stack allow five blocks per CU
#endif
// ALPHA: the scene has meshes with alpha masks (the mask test is compiled into this instance only)
// INST: the scene has object instances (TransformedPrimitive, primitive.cpp:78-99). A lane that meets an instance in a world
// leaf pushes ONE return entry (the rest of that leaf and the world ray's tMax; meta < 0 marks it), swaps its ray for
// Inverse(InstanceToWorld)(ray) and walks the object's tree above that entry; popping the entry reloads the world ray from
// the pool and resumes the leaf -- with the instance ray's tMax if something was hit inside (`r.tMax = ray.tMax`). The
// sequence of box tests, primitive tests and tMax updates per ray is the reference's recursion unrolled.
template <int MODE, bool ALPHA, int W, bool INST = false>
// MODE 3: the MIS rays again, as a visibility query (scenes without instances and without an alpha mask on an emitter's own mesh: DScene::misAny). The estimator
// reads one bit of a MIS ray -- is the closest hit the sampled light's shape (integrator.cpp:196-203) -- so k_shade hands
// over the span [tLo, tHi] of the ray in which that shape can be hit and the shape's primitive, and this walks the tree as the
// any-hit kernel does (no entry distances, no child ordering, five blocks per CU) with Triangle::Intersect's acceptance
// rules: a primitive other than the light's that is accepted below tLo ends the ray (occluded whatever the visiting order);
// one accepted inside the span only marks the ray ambiguous (the outcome depends on the order in which the reference meets
// the two), and an ambiguous ray, or one that met a quadric, is traversed again by the reference-order routine
// (k_resolve_overflow). Rays towards the environment light, and the rays whose answer nothing reads (F_MIS_DARK), carry
// tLo = tHi = infinity: any hit ends them.
#define TRAV_IS_ANY(MODE_) ((MODE_) == 1 || (MODE_) == 3)
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((TRAV_IS_ANY(MODE) && !ALPHA) ? MIPT_TRAV_WAVES_PER_EU_ANY : MIPT_TRAV_WAVES_PER_EU, (TRAV_IS_ANY(MODE) && !ALPHA) ? MIPT_TRAV_WAVES_PER_EU_ANY : MIPT_TRAV_WAVES_PER_EU)))
k_trav(DScene s, Pool pool, DevCounters *ctr) {
    constexpr bool ANY = TRAV_IS_ANY(MODE);
    constexpr bool PSEM = (MODE == 1);      // Triangle::IntersectP's acceptance rules (MODE 3 asks with Intersect's)
    constexpr bool MISANY = (MODE == 3);
    constexpr int QM = MISANY ? 2 : MODE;   // the queue and cursor of the class
    static_assert(!(MISANY && INST), "the visibility form of the MIS rays is not built for scenes with instances");
    int excl = -1;                          // MISANY: the sampled light's primitive
    float tLo = 0;                          // MISANY: hits accepted at or beyond it are ambiguous
    bool ambiguous = false;
    const int lane = threadIdx.x;
    const int wlane = threadIdx.x & 63;
    const unsigned nPrim = (MODE == 0) ? ctr->primCount.v : 0, nCont = (MODE == 0) ? ctr->contCount.v : 0;
    const unsigned total = (MODE == 0) ? nPrim + nCont : ((MODE == 1) ? ctr->shadowCount.v : ctr->misCount.v);
    const uint32_t *__restrict__ queue = (MODE == 0) ? pool.extQ : ((MODE == 1) ? pool.shadowQ : pool.misQ);
    (void)excl; (void)tLo; (void)ambiguous;
    const unsigned travChunk = (total >= (1u << 23)) ? 2u * (unsigned)TRAV_CHUNK : (unsigned)TRAV_CHUNK;
    const float4 *__restrict__ primTri = s.primTri;
    unsigned nodeCount = 0, triCount = 0, rayCount = 0;
    bool has = false;
    uint32_t slot = 0;
    unsigned myEntry = 0;   // MODE 1, 2: the ray's place in its queue (the answer goes beside the queue: shadowQ[n + myEntry], misQ[n + 2 myEntry])
    RayCtx r;
    InitRayCtx(r, 0, 0, 0, 1, 1, 1);
    float tMax = 0;
    TravState st;
    st.cur = -1; st.sp = 0;
    TravSpill spill;
    int nPend = 0, hitPrim = -1;
    float hitT = 0, hitB0 = 0, hitB1 = 0, hitB2 = 0;
    int leafOff = 0, leafCnt = 0;
    bool leafSimple = false;          // the leaf in hand holds triangles only (LEAF_SIMPLE)
    __shared__ int sTask[BLOCK];      // cooperative leaf test: per wave, task -> (owner lane, position in its leaf)
    int curInst = -1, hitInst = -1;   // INST: the instance the lane is inside, and the one its closest hit so far lies in
    bool hitInCur = false;            // INST: something was hit since the lane entered curInst
    TriRay triRay;
    triRay.kz = 2; triRay.Sx = triRay.Sy = 0; triRay.Sz = 1;
    bool exhausted = false;
#ifdef MIPT_TRAV_STATS
    unsigned long long dbgPasses = 0, dbgHas = 0, dbgWalk = 0, dbgWalkPasses = 0, dbgTriPasses = 0, dbgTri = 0;
#endif
    unsigned chunkNext = 0, chunkEnd = 0;  // wave-uniform: the range of the work list this wave reserved
    while (true) {
        // ---- fetch rays for idle lanes
        if (!exhausted) {
            const unsigned long long idle = __ballot(!has);
            if (idle) {
                if (chunkNext == chunkEnd) {  // the wave's private range is used up: reserve a chunk more
                    unsigned base = 0;
                    if (wlane == 0) base = atomicAdd(&ctr->travNext[QM].v, travChunk);
                    base = __shfl(base, 0, 64);
                    chunkNext = min(base, total);
                    chunkEnd = min(base + travChunk, total);
                    if (base >= total) exhausted = true;
                }
                const unsigned first = chunkNext;
                chunkNext = min(chunkNext + (unsigned)__popcll(idle), chunkEnd);
                if (!has) {
                    const unsigned my = first + __popcll(idle & ((1ull << wlane) - 1));
                    if (my < chunkEnd) {
                        if (MODE == 0) slot = queue[(my < nPrim) ? my : pool.n - nCont + (my - nPrim)];
                        else slot = queue[my];
                        if (MODE != 0) myEntry = my;
                        {
                            const float4 r0 = pool.R((MODE == 0) ? R_RAY0 : ((MODE == 1) ? R_SH0 : R_MI0), slot);
                            const float4 r1 = pool.R((MODE == 0) ? R_RAY1 : ((MODE == 1) ? R_SH1 : R_MI1), slot);
                            if (MODE == 0) InitRayCtx(r, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z);
                            else InitRayCtx(r, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y);
                            tMax = (MODE == 0) ? r0.w : ((MODE == 1) ? 1 - kShadowEpsilon : (MISANY ? r1.z : kInfinity));
                            if (MISANY) {   // (k_shade: R_MI1 = d.y, d.z, tHi, light primitive | width code << 27; tLo = tHi (1 - 2^-code))
                                const unsigned xw = __float_as_uint(r1.w);
                                excl = (int)(xw & MIS_EXCL_NONE);
                                const unsigned code = xw >> MIS_EXCL_BITS;
                                tLo = code == 0 ? 0.f : tMax * (1.f - __uint_as_float((127u - code) << 23));
                                ambiguous = false;
                            }
                            StartTraversal(s, r, tMax, st, nodeCount);
                            triRay = MakeTriRay(V3(r.dx, r.dy, r.dz));
                            nPend = 0; hitPrim = -1;
                            hitT = hitB0 = hitB1 = hitB2 = 0;
                            leafCnt = 0;
                            curInst = hitInst = -1; hitInCur = false;
                            ++rayCount;
                            if (st.cur >= 0) has = true;
                            else if (MODE == 1) pool.shadowQ[pool.n + myEntry] = 0u;   // nothing to traverse: unoccluded
                            else if (MODE == 2) {
                                *reinterpret_cast<uint2 *>(pool.misQ + pool.n + 2 * (size_t)myEntry) = make_uint2(0xffffffffu, 0u);
                                pool.R(R_HIT, slot) = make_float4(0.f, 0.f, 0.f, 0.f);
                            } else if (MISANY) {
                                *reinterpret_cast<uint2 *>(pool.misQ + pool.n + 2 * (size_t)myEntry) = make_uint2(0xffffffffu, 0u);
                            } else {  // the ray misses the world bound: nothing to traverse
                                pool.I(I_HITPRIM, slot) = -1;
                                pool.I(I_NPEND, slot) = 0;
                                if (INST && MODE == 0) pool.I(I_HITINST, slot) = -1;
                                if (!ANY) pool.R(R_HIT, slot) = make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                        }
                    }
                }
            }
        }
        if (!__any(has)) {
            if (exhausted) break;
            continue;  // every fetched slot was dead: fetch again
        }
        // ---- step the lanes until enough of them have run dry. Every pass opens one
        // interior node in each lane that holds one; lanes that reached a leaf wait, and once
        // TRI_BATCH lanes wait (or nobody is left walking) their leaves are tested -- triangles-only
        // leaves by the whole wave (cooperative test below), the others one primitive per waiting
        // lane -- so both the box code and the triangle code run on well-filled waves.
        // Per lane the sequence of node visits, pops and primitive tests is unchanged.
        while (true) {
            const bool walking = has && st.cur >= 0;
            const bool inLeaf = has && leafCnt > 0;
            const int nLeaf = __popcll(__ballot(inLeaf));
            const bool anyWalk = __any(walking);
#ifdef MIPT_TRAV_STATS
            ++dbgPasses; dbgHas += __popcll(__ballot(has)); dbgWalk += __popcll(__ballot(walking));
            if (anyWalk) ++dbgWalkPasses;
            if (nLeaf > 0 && (nLeaf >= TRI_BATCH || !anyWalk)) { ++dbgTriPasses; dbgTri += nLeaf; }
#endif
            bool needPop = false, got = false, finished = false;
            int tkChild = 0, tkMeta = 0;
            if (walking) {
#ifdef MIPT_NO_FAST_BOX
                got = OpenNode<W, !ANY>(s.wnodes, st.cur, r, tMax, spill, lane, st.sp, &tkChild, &tkMeta, nodeCount);
#else
                if (__any(r.slow)) got = OpenNode<W, !ANY, false>(s.wnodes, st.cur, r, tMax, spill, lane, st.sp, &tkChild, &tkMeta, nodeCount);
                else got = OpenNode<W, !ANY, W == 4>(s.wnodes, st.cur, r, tMax, spill, lane, st.sp, &tkChild, &tkMeta, nodeCount);
#endif
                needPop = !got;
                st.cur = -1;
            }
            const bool triPass = nLeaf > 0 && (nLeaf >= TRI_BATCH || !anyWalk);   // (wave-uniform)
            // ---- cooperative leaf test. A lane waiting at a triangles-only leaf hands up to COOP_KMAX of its (ray, triangle)
            // pairs to the wave: the pairs of all waiting lanes are numbered owner by owner, lane w tests pair w with the
            // owner's ray against the owner's tMax of this moment, and the owner then takes the outcome the sequential loop
            // of BVHAccel::Intersect would have reached: of the pairs that hit, in leaf order, each is accepted unless the
            // test's one tMax-dependent line (tScaled against tMax * det, triangle.cpp:262-266) rejects it with the tMax
            // left by the hit accepted before it. One pass finishes a whole leaf at the utilisation of (pairs / 64) instead
            // of one triangle per waiting lane (a fifth of the wave on the 10M-triangle scene).
            if (triPass && __any(inLeaf && leafSimple)) {
                const bool coop = inLeaf && leafSimple;
                const int want = coop ? min(leafCnt, COOP_KMAX) : 0;
                int base = 0, total = 0;
#pragma unroll
                for (int b = 0; b < COOP_KMAX; ++b) {
                    const unsigned long long m = __ballot(want > b);
                    base += __popcll(m & ((1ull << wlane) - 1));
                    total += __popcll(m);
                }
                const int grant = max(0, min(want, 64 - base));
                int *taskOwner = &sTask[threadIdx.x & ~63];
                for (int k = 0; k < grant; ++k) taskOwner[base + k] = wlane | (k << 8);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int nTask = min(total, 64);
                const bool worker = wlane < nTask;
                const int ow = worker ? taskOwner[wlane] : wlane;
                const int owner = ow & 63, tk = ow >> 8;
                const int wOff = __shfl(leafOff, owner, 64);
                const float wox = __shfl(r.ox, owner, 64), woy = __shfl(r.oy, owner, 64), woz = __shfl(r.oz, owner, 64);
                TriRay wtr;
                wtr.kz = __shfl(triRay.kz, owner, 64);
                wtr.Sx = __shfl(triRay.Sx, owner, 64); wtr.Sy = __shfl(triRay.Sy, owner, 64); wtr.Sz = __shfl(triRay.Sz, owner, 64);
                const float wT = __shfl(tMax, owner, 64);
                const int wExcl = MISANY ? __shfl(excl, owner, 64) : -1;
                const float wLo = MISANY ? __shfl(tLo, owner, 64) : 0.f;
                bool res = false, resClear = false;
                float rT = 0, rB0 = 0, rB1 = 0, rB2 = 0, rDet = 0, rTs = 0;
                if (worker) {
                    const int prim = wOff + tk;
                    const float4 v0 = primTri[3 * prim], v1 = primTri[3 * prim + 1], v2 = primTri[3 * prim + 2];
                    const unsigned pf = __float_as_uint(v0.w);
                    TriHit th;
                    if (TriTestRay(V3(v0.x, v0.y, v0.z), V3(v1.x, v1.y, v1.z), V3(v2.x, v2.y, v2.z), V3(wox, woy, woz), wtr, wT, &th, &rDet, &rTs)) {
                        bool counts = true;
                        if constexpr (ALPHA)
                            if (pf & PRIM_FLAG_ALPHA)
                                counts = !(pf & PRIM_FLAG_DEGENERATE) && AlphaPass(s, __float_as_int(v1.w), th.b0, th.b1, th.b2, PSEM);   // (the shadow mask is IntersectP's alone, triangle.cpp:531-570)
                        res = counts && (PSEM || !(pf & PRIM_FLAG_DEGENERATE));
                        if (MISANY) { res = res && prim != wExcl; resClear = res && th.t < wLo; }
                        rT = th.t; rB0 = th.b0; rB1 = th.b1; rB2 = th.b2;
                    }
                }
                const unsigned long long hitMask = __ballot(res);
                const unsigned seg = grant > 0 ? (unsigned)(hitMask >> base) & ((1u << grant) - 1u) : 0u;   // bit k: my k-th pair hit
                if (MISANY) {
                    const unsigned long long clearMask = __ballot(resClear);
                    const unsigned segClear = grant > 0 ? (unsigned)(clearMask >> base) & ((1u << grant) - 1u) : 0u;
                    if (segClear) { hitPrim = leafOff + (__ffs(segClear) - 1); finished = true; triCount += (unsigned)__ffs(segClear); }
                    else {
#ifndef MIPT_EXP_IGNORE_AMBIGUOUS   // (mutation build: tests/test_gpu_parity.py::test_mis_rays_... must fail with it)
                        if (seg) ambiguous = true;
#endif
                        triCount += (unsigned)grant;
                    }
                } else if (ANY) {
                    if (seg) { hitPrim = leafOff + (__ffs(seg) - 1); finished = true; triCount += (unsigned)__ffs(seg); }
                    else triCount += (unsigned)grant;
                } else {
                    int win = -1;
                    const bool multi = __popc(seg) >= 2;
                    if (__any(multi)) {   // two hits in one leaf: replay the sequence exactly
                        unsigned rem = multi ? seg : 0u;
                        float cur = tMax;
                        while (__any(rem != 0u)) {
                            const int k = rem ? (__ffs(rem) - 1) : 0;
                            const int src = rem ? base + k : wlane;
                            const float dk = __shfl(rDet, src, 64), tsk = __shfl(rTs, src, 64), tk2 = __shfl(rT, src, 64);
                            if (rem) {
                                const bool beyond = dk < 0 ? (tsk < cur * dk) : (tsk > cur * dk);
                                if (!beyond) { cur = tk2; win = k; }
                                rem &= rem - 1u;
                            }
                        }
                    }
                    if (!multi && seg) win = __ffs(seg) - 1;
                    const int src = win >= 0 ? base + win : wlane;
                    const float hT = __shfl(rT, src, 64), hB0 = __shfl(rB0, src, 64), hB1 = __shfl(rB1, src, 64), hB2 = __shfl(rB2, src, 64);
                    if (win >= 0) {
                        tMax = hT;
                        hitPrim = leafOff + win; hitT = hT; hitB0 = hB0; hitB1 = hB1; hitB2 = hB2;
                        if (INST) { hitInst = curInst; hitInCur = true; }
                    }
                    triCount += (unsigned)grant;
                }
                if (grant > 0) {
                    leafOff += grant; leafCnt -= grant;
                    needPop = !finished && leafCnt == 0;
                }
            }
            if (triPass && inLeaf && !leafSimple) {
                const int prim = leafOff;
                ++leafOff; --leafCnt;
                const float4 v0 = primTri[3 * prim];
                const unsigned pf = __float_as_uint(v0.w);
                bool entered = false;
                if (INST && (pf & PRIM_FLAG_INSTANCE)) {
                    // TransformedPrimitive::Intersect[P]: the ray in the instance's space, then the object's BVH from its root
                    const int k = __float_as_int(primTri[3 * prim + 1].w);
                    StackPush<!ANY>(spill, lane, st.sp, leafOff, -(leafCnt + 1), tMax);
                    const Ray ir = XfRay(s.instances[k].w2i, Ray(V3(r.ox, r.oy, r.oz), V3(r.dx, r.dy, r.dz), tMax));
                    InitRayCtx(r, ir.o.x, ir.o.y, ir.o.z, ir.d.x, ir.d.y, ir.d.z);
                    triRay = MakeTriRay(ir.d);
                    tMax = ir.tMax;
                    curInst = k; hitInCur = false;
                    leafCnt = 0;
                    const float4 bMin = s.instRootBounds[2 * k], bMax = s.instRootBounds[2 * k + 1];
                    float tBox;
                    ++nodeCount;
                    got = BoxTest(r, bMin.x, bMin.y, bMin.z, bMax.x, bMax.y, bMax.z, tMax, &tBox);
                    tkChild = s.instWideRoot[k]; tkMeta = 0;
                    entered = true;
                } else if (INST && curInst >= 0 && (pf & PRIM_FLAG_SPHERE)) {
                    // a quadric of an instanced object: tested where it is met, with the instance's ray
                    float t;
                    if (SphereHitT(s.spheres[__float_as_int(primTri[3 * prim + 1].w)], V3(r.ox, r.oy, r.oz), V3(r.dx, r.dy, r.dz), tMax, &t)) {
                        if (ANY) { hitPrim = prim; finished = true; }
                        else { tMax = t; hitPrim = prim; hitT = t; hitB0 = hitB1 = hitB2 = 0; hitInst = curInst; hitInCur = true; }
                    }
                } else
                if (MISANY && prim == excl) {}   // the sampled light's own shape
                else if (pf & PRIM_FLAG_SPHERE) {
                    if ((nPend & 0xff) < MAX_PEND) {
                        pool.I(I_PEND0 + (nPend & 0xff), slot) = prim;
                        if (!ANY) pool.F(P_TENC0 + (nPend & 0xff), slot) = tMax;   // (see ResolveQuadrics)
                        ++nPend;
                    }
                    else nPend |= PEND_OVERFLOW;
                } else {
                    const float4 v1 = primTri[3 * prim + 1], v2 = primTri[3 * prim + 2];
                    ++triCount;
                    TriHit th;
                    if (TriTestRay(V3(v0.x, v0.y, v0.z), V3(v1.x, v1.y, v1.z), V3(v2.x, v2.y, v2.z), V3(r.ox, r.oy, r.oz),
                                   triRay, tMax, &th)) {
                        // meshes with an alpha mask: IntersectP then rejects degenerate triangles too, and both reject
                        // hits where the mask is 0 (triangle.cpp:331-338, 531-570)
                        bool counts = true;
                        if constexpr (ALPHA)
                            if (pf & PRIM_FLAG_ALPHA)
                                counts = !(pf & PRIM_FLAG_DEGENERATE) && AlphaPass(s, __float_as_int(v1.w), th.b0, th.b1, th.b2, PSEM);   // (the shadow mask is IntersectP's alone, triangle.cpp:531-570)
                        if (!counts) {}
                        else if (PSEM) { hitPrim = prim; finished = true; }
                        else if (MISANY) {
                            if (!(pf & PRIM_FLAG_DEGENERATE)) {
                                if (th.t < tLo) { hitPrim = prim; finished = true; }
                                else ambiguous = true;
                            }
                        }
                        else if (!(pf & PRIM_FLAG_DEGENERATE)) {
                            tMax = th.t;
                            hitPrim = prim; hitT = th.t; hitB0 = th.b0; hitB1 = th.b1; hitB2 = th.b2;
                            if (INST) { hitInst = curInst; hitInCur = true; }
                        }
                    }
                }
                needPop = entered ? !got : (!finished && leafCnt == 0);
            }
            if (needPop) {  // a popped node is entered only if still in front of tMax
                while (!got && st.sp > 0) {
                    float t;
                    StackPop<!ANY>(spill, lane, st.sp, &tkChild, &tkMeta, &t);
                    if (INST && tkMeta < 0) {   // back from the instance: the world ray again, and the rest of the leaf
                        const float4 r0 = pool.R((MODE == 0) ? R_RAY0 : ((MODE == 1) ? R_SH0 : R_MI0), slot);
                        const float4 r1 = pool.R((MODE == 0) ? R_RAY1 : ((MODE == 1) ? R_SH1 : R_MI1), slot);
                        if (MODE == 0) InitRayCtx(r, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z);
                        else InitRayCtx(r, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y);
                        triRay = MakeTriRay(V3(r.dx, r.dy, r.dz));
                        if (!hitInCur) tMax = ANY ? 1 - kShadowEpsilon : t;   // (a hit inside: r.tMax = ray.tMax; any-hit: no entry distances are kept)
                        curInst = -1;
                        tkMeta = -tkMeta - 1;      // primitives left in the world leaf
                        got = tkMeta > 0;
                        continue;
                    }
                    got = ANY || t < tMax;
                }
                finished = !got;
            }
            if (got) {
                if (tkMeta > 0) { leafOff = tkChild; leafCnt = tkMeta & LEAF_COUNT_MASK; leafSimple = (tkMeta & LEAF_SIMPLE) != 0; }
                else st.cur = tkChild;
            }
            if (finished && MODE == 1) {
                pool.shadowQ[pool.n + myEntry] = (hitPrim >= 0 ? 0x80000000u : 0u) | (unsigned)nPend;
                has = false;
                leafCnt = 0;
            } else
            if (finished && MODE == 2) {
                *reinterpret_cast<uint2 *>(pool.misQ + pool.n + 2 * (size_t)myEntry) = make_uint2((unsigned)hitPrim, (unsigned)nPend);
                pool.R(R_HIT, slot) = make_float4(hitT, hitB0, hitB1, hitB2);
                has = false;
                leafCnt = 0;
            } else
            if (finished && MISANY) {   // an occluder's primitive, -1: nothing in front of the light's span, -2: ambiguous
                *reinterpret_cast<uint2 *>(pool.misQ + pool.n + 2 * (size_t)myEntry) = make_uint2(hitPrim >= 0 ? (unsigned)hitPrim : (ambiguous ? 0xfffffffeu : 0xffffffffu), (unsigned)nPend);
                has = false;
                leafCnt = 0;
            } else
            if (finished) {
                pool.I(I_HITPRIM, slot) = hitPrim;
                pool.I(I_NPEND, slot) = nPend;
                if (!ANY) pool.R(R_HIT, slot) = make_float4(hitT, hitB0, hitB1, hitB2);
                if (INST && MODE == 0) pool.I(I_HITINST, slot) = hitInst;
                has = false;
                leafCnt = 0;
            }
            const int active = __popcll(__ballot(has));
            if (active == 0 || (!exhausted && active < REFILL_BELOW)) break;
        }
    }
#ifdef MIPT_TRAV_STATS
    if (blockIdx.x == 7 && threadIdx.x == 0 && total > 1000000)
        printf("TRAVSTAT mode %d total %u passes %llu has/pass %.1f walkPasses %llu walk/walkPass %.1f triPasses %llu tri/triPass %.1f\n", MODE, total, dbgPasses,
               (double)dbgHas / dbgPasses, dbgWalkPasses, (double)dbgWalk / (dbgWalkPasses ? dbgWalkPasses : 1), dbgTriPasses, (double)dbgTri / (dbgTriPasses ? dbgTriPasses : 1));
#endif
    DevStats &st8 = Stats(ctr);
    if (MODE == 1) CountAdd(&st8.shadowRays, rayCount);
    else CountAdd(&st8.regularRays, rayCount);
    CountAdd(&st8.nodesVisited, nodeCount);
    CountAdd(&st8.triTests, triCount);
    if (MODE == 0) { CountAdd(&st8.extendNodes, nodeCount); CountAdd(&st8.extendTris, triCount); CountAdd(&st8.extendRays, rayCount); }
}

// Quadrics recorded by k_trav, tested after the triangles at full lane utilisation -- with the outcome of the reference's
// order. Sphere::Intersect rejects a root whose error-bounded UPPER end lies beyond tMax (sphere.cpp:77-82), so when a
// triangle and a sphere are met within that error bound of each other (a sphere resting on a quad: fuzz scene 2004), which
// one is the hit depends on which was met first. So each postponed quadric keeps the ray's tMax of the moment it was met
// (P_TENC*), and this replays the reference's sequence: quadric j is tested against min(that tMax, the closest quadric
// accepted before it); a triangle hit that the traversal found AFTER the last accepted quadric survives only if it is
// closer than that quadric. (Shadow rays: tMax never changes, the order does not matter.) On overflow the ray is
// re-traversed by the reference-order routine with inline quadric tests.
// OVF: the instance of k_resolve_overflow. The resolve kernels proper are compiled without the re-traversal (it cost
// them 40-56 VGPRs and all but 650 of 12 000 instructions): they hand a ray whose list overflowed to that kernel.
template <bool ANY, bool INST, bool OVF = false>
DEV bool ResolveQuadrics(const DScene &s, const Pool &pool, uint32_t slot, const V3 &ro, const V3 &rd, float tMaxIn,
                         Hit *h, bool foundTri, unsigned &nodes, unsigned &tris, int npGiven = -1) {
    const int np = npGiven >= 0 ? npGiven : pool.I(I_NPEND, slot);
    if constexpr (OVF) {
        Hit h2;
        h2.prim = -1; h2.t = 0; h2.b0 = h2.b1 = h2.b2 = 0;
        unsigned n2 = 0, t2 = 0;  // statistics were already counted by k_trav
        const bool found = Traverse<ANY, INST>(s, ro, rd, tMaxIn, &h2, n2, t2);
        if (found) *h = h2;
        return found;
    }
    if (ANY) {
        for (int j = 0; j < (np & 0xff); ++j) {
            const int prim = pool.I(I_PEND0 + j, slot);
            float t;
            if (SphereHitT(s.spheres[__float_as_int(s.primTri[3 * prim + 1].w)], ro, rd, tMaxIn, &t)) return true;
        }
        return foundTri;
    }
    float tBest = kInfinity, tEncBest = 0.f;
    int primBest = -1;
    for (int j = 0; j < (np & 0xff); ++j) {
        const int prim = pool.I(I_PEND0 + j, slot);
        const float tEnc = pool.F(P_TENC0 + j, slot);
        float t;
        if (SphereHitT(s.spheres[__float_as_int(s.primTri[3 * prim + 1].w)], ro, rd, minf(tEnc, tBest), &t)) { tBest = t; primBest = prim; tEncBest = tEnc; }
    }
    if (primBest < 0) return foundTri;
    if (foundTri && tEncBest > h->t && h->t <= tBest) return true;   // the triangle was found later, and in front of the quadric
    h->prim = primBest; h->t = tBest; h->b0 = h->b1 = h->b2 = 0; h->inst = -1;
    return true;
}

// Slots per block of the kernels that walk the whole pool (k_generate, k_resolve_extend): SLOT_CHUNKS x 256. Every block
// ends in a returning atomicAdd on a queue cursor, and one word takes ~88 of those per microsecond whoever issues them
// (MI355X_MICROARCH.md, "dequeue"): with one 256-slot chunk per block a 32M-slot pool sent 131k adds to each cursor per
// launch, a floor of 1.5 ms under kernels that have 1.4-2 ms of work. Eight chunks per block: 16k adds, 0.19 ms.
#ifndef MIPT_SLOT_CHUNKS
#define MIPT_SLOT_CHUNKS 8
#endif
constexpr int SLOT_CHUNKS = MIPT_SLOT_CHUNKS;
// k_resolve_extend keeps a slot's rank within its (chunk, wave, class) in 8 bits per chunk of two 32-bit words (rankLo /
// rankHi), k_generate a bit per chunk in 32-bit masks and the block's slot numbers in 16 bits: more than 8 chunks would
// shift ranks out of their word (round 2's `-DMIPT_SLOT_CHUNKS=16` tuning build queued slots twice, overran a ray queue and
// died of a GPU memory fault -- the "Aborted" of gpurun_out/call_var.log -- which also left the device unusable for the two
// builds benched after it in that call). The tuning macros are checked where they are defined.
static_assert(SLOT_CHUNKS >= 1 && SLOT_CHUNKS <= 8, "MIPT_SLOT_CHUNKS: 1..8 (rank words of k_resolve_extend, chunk masks of k_generate)");
static_assert(SLOT_CHUNKS * BLOCK / 2 + SLOT_CHUNKS * BLOCK / 4 <= BLOCK * 33, "k_generate: the free-slot list and its flags live in the flush rows");

template <bool INST>
__global__ void __launch_bounds__(BLOCK) k_resolve_extend(DScene s, Pool pool, DevCounters *ctr) {
    // append to the class queues: one atomic per class and block
    __shared__ unsigned sCnt[SLOT_CHUNKS][BLOCK / 64][MAX_CLASSES], sBase[MAX_CLASSES];
    // The rays with postponed quadrics are a minority spread over the block's slots: they are listed first and resolved
    // afterwards by all lanes together, so the interval-arithmetic sphere test runs on full waves instead of on the few
    // lanes of each chunk that hold such a ray (Cornell: a seventh of the rays, the kernel 2.5x faster).
    __shared__ unsigned char sCls[SLOT_CHUNKS][BLOCK];   // per slot of the block: 0x80 | class when the slot holds a traced path
    __shared__ unsigned short sPend[SLOT_CHUNKS * BLOCK];
    __shared__ unsigned sPendCount;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned nodes = 0, tris = 0;
    if (threadIdx.x == 0) sPendCount = 0;
    __syncthreads();
    const float4 *__restrict__ primTri = s.primTri;   // (by value: a reference to the kernel argument would move it to scratch)
    auto classOf = [primTri](int prim) -> int {
        return (prim >= 0) ? (int)((__float_as_uint(primTri[3 * prim].w) >> PRIM_CLASS_SHIFT) & (unsigned)(MAX_CLASSES - 1)) : MISS_CLASS;
    };
    // ---- phase A: every slot's hit primitive; rays with a quadric list go onto the block's list
#pragma unroll 1
    for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {
        const uint32_t slot = (blockIdx.x * SLOT_CHUNKS + ch) * BLOCK + threadIdx.x;
        unsigned char cc = 0;
        if (slot < pool.n && (pool.I(I_FLAGS, slot) & F_ALIVE)) {
            const int npend = pool.I(I_NPEND, slot);
            if (npend & PEND_OVERFLOW) pool.ovfQ[atomicAdd(&ctr->ovfCount[0].v, 1u)] = slot;   // k_resolve_overflow commits this one
            else if (npend != 0) sPend[atomicAdd(&sPendCount, 1u)] = (unsigned short)(ch * BLOCK + threadIdx.x);
            else cc = (unsigned char)(0x80 | classOf(pool.I(I_HITPRIM, slot)));
        }
        sCls[ch][threadIdx.x] = cc;
    }
    __syncthreads();
    // ---- phase B: the listed rays, one per lane
    const unsigned nPend = sPendCount;
#pragma unroll 1
    for (unsigned i = threadIdx.x; i < nPend; i += BLOCK) {
        const unsigned e = sPend[i];
        const uint32_t slot = blockIdx.x * SLOT_CHUNKS * BLOCK + e;
        int prim = pool.I(I_HITPRIM, slot);
        const float4 r0 = pool.R(R_RAY0, slot), r1 = pool.R(R_RAY1, slot), hr = pool.R(R_HIT, slot);
        V3 ro(r0.x, r0.y, r0.z), rd(r1.x, r1.y, r1.z);
        Hit h;
        h.prim = prim; h.t = hr.x; h.b0 = hr.y; h.b1 = hr.z; h.b2 = hr.w;
        if (INST) h.inst = pool.I(I_HITINST, slot);
        const bool found = ResolveQuadrics<false, INST>(s, pool, slot, ro, rd, r0.w, &h, prim >= 0, nodes, tris);
        prim = found ? h.prim : -1;
        pool.I(I_HITPRIM, slot) = prim;
        pool.R(R_HIT, slot) = make_float4(h.t, h.b0, h.b1, h.b2);
        if (INST) pool.I(I_HITINST, slot) = h.inst;
        sCls[e / BLOCK][e % BLOCK] = (unsigned char)(0x80 | classOf(prim));
    }
    __syncthreads();
    // ---- phase C: ranks within (chunk, wave, class), the block's range of each class queue, the queue entries
    unsigned rankLo = 0, rankHi = 0;              // per chunk: 8 bits of rank within the wave and class
#pragma unroll 1
    for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {
        const unsigned cc = sCls[ch][threadIdx.x];
        const bool traced = (cc & 0x80u) != 0;
        const int cls = (int)(cc & 0x7fu);
        unsigned rank = 0;
        for (int c = 0; c < MAX_CLASSES; ++c) {
            if (!((s.classMask >> c) & 1)) continue;
            const unsigned long long m = __ballot(traced && cls == c);
            if (lane == 0) sCnt[ch][wave][c] = (unsigned)__popcll(m);
            if (traced && cls == c) rank = (unsigned)__popcll(m & ((1ull << lane) - 1));
        }
        if (ch < 4) rankLo |= rank << (8 * ch); else rankHi |= rank << (8 * (ch - 4));
    }
    __syncthreads();
    if (threadIdx.x < MAX_CLASSES && ((s.classMask >> threadIdx.x) & 1)) {
        const int c = threadIdx.x;   // exclusive prefix over (chunk, wave), then the block's range of queue c in one add
        unsigned tot = 0;
        for (int ch = 0; ch < SLOT_CHUNKS; ++ch)
            for (int w = 0; w < BLOCK / 64; ++w) { const unsigned n = sCnt[ch][w][c]; sCnt[ch][w][c] = tot; tot += n; }
        sBase[c] = tot ? atomicAdd(&ctr->shadeCount[c].v, tot) : 0;
    }
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {
        const unsigned cc = sCls[ch][threadIdx.x];
        if (!(cc & 0x80u)) continue;
        const int cls = (int)(cc & 0x7fu);
        const unsigned rank = ((ch < 4 ? rankLo >> (8 * ch) : rankHi >> (8 * (ch - 4))) & 255u);
        const uint32_t slot = (blockIdx.x * SLOT_CHUNKS + ch) * BLOCK + threadIdx.x;
        pool.shadeQ[(size_t)cls * pool.n + sBase[cls] + sCnt[ch][wave][cls] + rank] = slot;
    }
}

template <bool INST>
__global__ void __launch_bounds__(BLOCK) k_resolve_shadow(DScene s, Pool pool, DevCounters *ctr) {
    const uint32_t qi = blockIdx.x * BLOCK + threadIdx.x;
    unsigned zero = 0, nodes = 0, tris = 0;
    int myFlags = 0;
    uint32_t mySlot = 0;
    bool valid = false, doAdd = false;
    if (qi < ctr->shadowCount.v) {
        const uint32_t slot = pool.shadowQ[qi];
        int flags = pool.I(I_FLAGS, slot);
        const unsigned verdict = pool.shadowQ[pool.n + qi];   // k_trav<1>'s answer, in queue order
        bool occluded = (verdict >> 31) != 0u;
        const int npend = occluded ? 0 : (int)(verdict & 0x7fffffffu);
        if (npend & PEND_OVERFLOW) pool.ovfQ[(size_t)pool.n + atomicAdd(&ctr->ovfCount[1].v, 1u)] = slot;   // k_resolve_overflow commits this one
        else {
            if (npend != 0) {
                const float4 r0 = pool.R(R_SH0, slot), r1 = pool.R(R_SH1, slot);
                V3 ro(r0.x, r0.y, r0.z), rd(r0.w, r1.x, r1.y);
                Hit h;
                occluded = ResolveQuadrics<true, INST>(s, pool, slot, ro, rd, 1 - kShadowEpsilon, &h, false, nodes, tris, npend);
            }
            myFlags = flags; mySlot = slot; valid = true; doAdd = !occluded;
        }
    }
    // L += contribution: the candidate line holds the sum already (F_CAND, written by k_shade), an unoccluded ray makes it the
    // path's L; an occluded one leaves L where it is
    if (valid) {
        int flags = myFlags;
        const bool added = doAdd && (flags & F_NEE_NZ);
        if (doAdd) flags = (flags ^ F_L_IN_B) & ~F_L_ZERO;
        flags &= ~(F_SHADOW | F_CAND | F_NEE_NZ);
        if (flags & F_MIS) { if (added) flags |= F_A_ADDED; }   // k_resolve_mis closes the estimate
        else { if (!added) ++zero; flags &= ~(F_NEE | F_A_ADDED); }
        pool.I(I_FLAGS, mySlot) = flags;
    }
    CountAdd(&Stats(ctr).zeroRadiancePaths, zero);
}


// One MIS ray's commit. OVF = false (k_resolve_mis): a ray whose quadric list overflowed goes to k_resolve_overflow, which
// runs this again with OVF = true.
template <bool INST, bool OVF>
DEV void ResolveMisSlot(const DScene &s, const Pool &pool, DevCounters *ctr, uint32_t slot, int hitPrim, int npend, unsigned &zero) {
    unsigned nodes = 0, tris = 0;
    int flags = pool.I(I_FLAGS, slot);
    // the ray and the hit record are fetched only by the few rays that need them: postponed quadrics, or a hit on the
    // sampled light whose facing has to be tested (most MIS rays hit something else: 48 B of scattered reads saved)
    V3 ro, rd;
    Hit h;
    h.prim = hitPrim; h.t = 0.f; h.b0 = h.b1 = h.b2 = 0.f;
    bool haveRay = false;
    auto loadRay = [&]() {
        if (haveRay) return;
        const float4 r0 = pool.R(R_MI0, slot), r1 = pool.R(R_MI1, slot), hr = pool.R(R_HIT, slot);
        ro = V3(r0.x, r0.y, r0.z); rd = V3(r0.w, r1.x, r1.y);
        if (h.prim >= 0) { h.t = hr.x; h.b0 = hr.y; h.b1 = hr.z; h.b2 = hr.w; }
        haveRay = true;
    };
    bool found = h.prim >= 0;
    if (!OVF && (npend & PEND_OVERFLOW)) { pool.ovfQ[2 * (size_t)pool.n + atomicAdd(&ctr->ovfCount[2].v, 1u)] = slot; return; }
    if (npend != 0) { loadRay(); found = ResolveQuadrics<false, INST, OVF>(s, pool, slot, ro, rd, kInfinity, &h, found, nodes, tris, npend); }
    bool added = false;
    const int misLight = s.nLights > 1 ? pool.I(I_MISLIGHT, slot) : 0;
    if (!found && s.lights[misLight].type == MI_LIGHT_INFINITE) {   // Li = light.Le(ray), integrator.cpp:204
        const bool lZero = (flags & F_L_ZERO) != 0;
        for (int c = 0; c < NQ; ++c) {
            const float4 a = pool.Q(Q_LMIS + c, slot);
            added |= (a.x != 0.f) | (a.y != 0.f) | (a.z != 0.f) | (a.w != 0.f);
            float4 l = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!lZero) l = pool.Q(LPlane(flags) + c, slot);
            l.x += a.x; l.y += a.y; l.z += a.z; l.w += a.w;
            pool.Q(LPlane(flags) + c, slot) = l;
        }
        flags &= ~F_L_ZERO;
    }
    if (found) {
        const int lightNum = misLight;
        if (s.prims[h.prim].area_light == lightNum && !(flags & F_MIS_DARK)) {   // (dark: cannot happen, the bounds are conservative)
            const mi_light &l = s.lights[lightNum];
            bool emit = l.two_sided != 0;
            if (!emit) {
                loadRay();
                SurfaceInteraction li;
                HitInteraction(s, h.prim, ro, rd, h.b0, h.b1, h.b2, &li);
                emit = Dot(li.n, -rd) > 0;
            }
            if (emit) {
                const bool lZero = (flags & F_L_ZERO) != 0;
                for (int c = 0; c < NQ; ++c) {
                    const float4 a = pool.Q(Q_LMIS + c, slot);
                    added |= (a.x != 0.f) | (a.y != 0.f) | (a.z != 0.f) | (a.w != 0.f);
                    float4 l = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!lZero) l = pool.Q(LPlane(flags) + c, slot);
                    l.x += a.x; l.y += a.y; l.z += a.z; l.w += a.w;
                    pool.Q(LPlane(flags) + c, slot) = l;
                }
                flags &= ~F_L_ZERO;
            }
        }
    }
    if (!added && !(flags & F_A_ADDED)) ++zero;
    pool.I(I_FLAGS, slot) = flags & ~(F_NEE | F_MIS | F_A_ADDED | F_MIS_DARK);
}
// The commit of a MIS ray that k_trav<3> answered as a visibility query (DScene::misAny). word0: an occluder's primitive,
// -1 (nothing accepted up to the end of the light's span) or -2 (something accepted inside the span). A ray that is
// ambiguous, met a quadric on its way to an area light, or reaches a sphere whose root the reference could reject against a
// hit beyond the span, goes to k_resolve_overflow: the closest-hit routine in the reference's order, then ResolveMisSlot.
DEV void ResolveMisVisibility(const DScene &s, const Pool &pool, DevCounters *ctr, uint32_t slot, int word0, int npend, unsigned &zero) {
    unsigned nodes = 0, tris = 0;
    int flags = pool.I(I_FLAGS, slot);
    const bool dark = (flags & F_MIS_DARK) != 0;   // (nothing reads the ray's answer)
    bool exact = !dark && (npend & PEND_OVERFLOW) != 0, add = false;
    if (!exact && !dark) {
        const int misLight = s.nLights > 1 ? pool.I(I_MISLIGHT, slot) : 0;
        const mi_light &l = s.lights[misLight];
        if (l.type == MI_LIGHT_INFINITE) {   // Li = light.Le(ray) if nothing is hit, integrator.cpp:204
            bool found = word0 >= 0;
            if (!found && (npend & 0xff)) {
                const float4 r0 = pool.R(R_MI0, slot), r1 = pool.R(R_MI1, slot);
                Hit h;
                found = ResolveQuadrics<true, false, false>(s, pool, slot, V3(r0.x, r0.y, r0.z), V3(r0.w, r1.x, r1.y), kInfinity, &h, false, nodes, tris, npend);
            }
            add = !found;
        } else if (word0 >= 0) {}           // a primitive in front of the light's span
        else if (word0 == -2 || (npend & 0xff)) exact = true;
        else {                              // nothing up to the end of the span: the light's shape, if the ray meets it, is the closest hit
            const float4 r0 = pool.R(R_MI0, slot), r1 = pool.R(R_MI1, slot);
            const V3 ro(r0.x, r0.y, r0.z), rd(r0.w, r1.x, r1.y);
            if (l.shape < 0) {
                const mi_sphere &sp = s.spheres[~l.shape];
                SurfaceInteraction li;
                float t, t2;
                if (SphereInteraction(sp, ro, rd, kInfinity, &li, &t)) {
                    if (!SphereHitT(sp, ro, rd, r1.z, &t2)) exact = true;   // (its upper error bound lies beyond the span)
                    else add = l.two_sided != 0 || Dot(li.n, -rd) > 0;
                }
            } else if (l.two_sided != 0) add = true;   // (k_shade: Shape::Pdf has intersected the triangle, else there is no ray)
            else {
                const int32_t *v = &s.triIndices[3 * l.shape];
                TriHit th;
                if (TriTest(LoadV3(s.P, v[0]), LoadV3(s.P, v[1]), LoadV3(s.P, v[2]), ro, rd, kInfinity, &th)) {
                    SurfaceInteraction li;
                    TriInteraction(s, l.shape, th.b0, th.b1, th.b2, rd, &li);
                    add = Dot(li.n, -rd) > 0;
                }
            }
        }
    }
    if (exact) { pool.ovfQ[2 * (size_t)pool.n + atomicAdd(&ctr->ovfCount[2].v, 1u)] = slot; return; }
    bool added = false;
    if (add) {
        const bool lZero = (flags & F_L_ZERO) != 0;
        for (int c = 0; c < NQ; ++c) {
            const float4 a = pool.Q(Q_LMIS + c, slot);
            added |= (a.x != 0.f) | (a.y != 0.f) | (a.z != 0.f) | (a.w != 0.f);
            float4 lq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!lZero) lq = pool.Q(LPlane(flags) + c, slot);
            lq.x += a.x; lq.y += a.y; lq.z += a.z; lq.w += a.w;
            pool.Q(LPlane(flags) + c, slot) = lq;
        }
        flags &= ~F_L_ZERO;
    }
    if (!added && !(flags & F_A_ADDED)) ++zero;
    pool.I(I_FLAGS, slot) = flags & ~(F_NEE | F_MIS | F_A_ADDED | F_MIS_DARK);
}
template <bool INST>
__global__ void __launch_bounds__(BLOCK) k_resolve_mis(DScene s, Pool pool, DevCounters *ctr) {
    const uint32_t qi = blockIdx.x * BLOCK + threadIdx.x;
    unsigned zero = 0;
    if (qi < ctr->misCount.v) {
        const uint2 v = *reinterpret_cast<const uint2 *>(pool.misQ + pool.n + 2 * (size_t)qi);   // k_trav<2>'s / k_trav<3>'s answer, in queue order
        if (!INST && s.misAny) ResolveMisVisibility(s, pool, ctr, pool.misQ[qi], (int)v.x, (int)v.y, zero);
        else ResolveMisSlot<INST, false>(s, pool, ctr, pool.misQ[qi], (int)v.x, (int)v.y, zero);
    }
    CountAdd(&Stats(ctr).zeroRadiancePaths, zero);
}

// The rays whose list of postponed quadrics overflowed (more than MAX_PEND quadrics met: PEND_OVERFLOW), handed over by
// the three resolve kernels: re-traversed by the reference-order routine with inline quadric tests, then committed as the
// resolve kernel would have (mode 0: hit record + shading queue; 1: occlusion + L += the light sample; 2: the MIS commit).
// A fixed small grid walks the queue; it is empty for all but quadric-heavy scenes.
constexpr int OVERFLOW_GRID = 512;
template <bool INST>
__global__ void __launch_bounds__(BLOCK) k_resolve_overflow(DScene s, Pool pool, DevCounters *ctr, int mode) {
    const unsigned count = ctr->ovfCount[mode].v;
    unsigned zero = 0, nodes = 0, tris = 0;
    for (unsigned qi = blockIdx.x * BLOCK + threadIdx.x; qi < count; qi += gridDim.x * BLOCK) {
        const uint32_t slot = pool.ovfQ[(size_t)mode * pool.n + qi];
        if (mode == 0) {
            const float4 r0 = pool.R(R_RAY0, slot), r1 = pool.R(R_RAY1, slot);
            Hit h;
            h.prim = -1; h.t = 0.f; h.b0 = h.b1 = h.b2 = 0.f;
            const bool found = ResolveQuadrics<false, INST, true>(s, pool, slot, V3(r0.x, r0.y, r0.z), V3(r1.x, r1.y, r1.z), r0.w, &h, false, nodes, tris);
            const int prim = found ? h.prim : -1;
            pool.I(I_HITPRIM, slot) = prim;
            pool.R(R_HIT, slot) = make_float4(h.t, h.b0, h.b1, h.b2);
            if (INST) pool.I(I_HITINST, slot) = h.inst;
            const int cls = (prim >= 0) ? (int)((__float_as_uint(s.primTri[3 * prim].w) >> PRIM_CLASS_SHIFT) & (unsigned)(MAX_CLASSES - 1)) : MISS_CLASS;
            pool.shadeQ[(size_t)cls * pool.n + atomicAdd(&ctr->shadeCount[cls].v, 1u)] = slot;
        } else if (mode == 1) {
            const float4 r0 = pool.R(R_SH0, slot), r1 = pool.R(R_SH1, slot);
            Hit h;
            const bool occluded = ResolveQuadrics<true, INST, true>(s, pool, slot, V3(r0.x, r0.y, r0.z), V3(r0.w, r1.x, r1.y), 1 - kShadowEpsilon, &h, false, nodes, tris);
            int flags = pool.I(I_FLAGS, slot);
            bool added = false;
            if (!occluded) { added = (flags & F_NEE_NZ) != 0; flags = (flags ^ F_L_IN_B) & ~F_L_ZERO; }   // the candidate becomes L (k_resolve_shadow)
            flags &= ~(F_SHADOW | F_CAND | F_NEE_NZ);
            if (flags & F_MIS) { if (added) flags |= F_A_ADDED; }
            else { if (!added) ++zero; flags &= ~(F_NEE | F_A_ADDED); }
            pool.I(I_FLAGS, slot) = flags;
        } else
            ResolveMisSlot<INST, true>(s, pool, ctr, slot, -1, PEND_OVERFLOW, zero);   // (re-traversed from scratch)
    }
    CountAdd(&Stats(ctr).zeroRadiancePaths, zero);
}

// ------------------------------------------------------------------ generate
DEV void CameraRay(const DScene &s, float pFilmX, float pFilmY, float lensU, float lensV, Ray *out) {
    // PerspectiveCamera::GenerateRayDifferential, perspective.cpp:95-146 (differentials feed
    // only texture filtering; every texture on this path is constant)
    const mi_camera &cam = s.camera;
    V3 pCamera = XfPoint(cam.raster_to_camera, V3(pFilmX, pFilmY, 0));
    V3 dir = Normalize(V3(pCamera.x, pCamera.y, pCamera.z));
    Ray ray(V3(0, 0, 0), dir);
    if (cam.lens_radius > 0) {
        float dx, dy;
        ConcentricSampleDisk(lensU, lensV, &dx, &dy);
        float lx = cam.lens_radius * dx, ly = cam.lens_radius * dy;
        float ft = cam.focal_distance / ray.d.z;
        V3 pFocus = ray.at(ft);
        ray.o = V3(lx, ly, 0);
        ray.d = Normalize(pFocus - ray.o);
    }
    *out = XfRay(cam.camera_to_world, ray);
}

#if MIPT_HAS_MAIN
// FilmTile::AddSample + MergeFilmTile (film.h:123-163, film.cpp:124-142) as float atomics
// into the resident film [pixel][32] (31 bins + filter-weight sum = one 128-B row), then the refill of the
// freed slots with new camera samples, then the extend work list.
//
// A block owns SLOT_CHUNKS x 256 consecutive slots and makes three passes over them, so that everything it needs from a
// shared cursor is asked for ONCE per block (a word takes ~88 returning atomics per microsecond; see SLOT_CHUNKS):
//   pass 1  flush the finished paths of each chunk into the film -- each finished lane stages its guarded radiance in LDS
//           (row stride 33 words: conflict-free both ways); the samples of a wave that go to the same pixel are summed
//           there and leave as ONE 128-B row update, two pixels per wave instruction (the full-rate shape for gfx950
//           float atomics) -- and count the slots that are free now;
//   ------  one atomicAdd on the work counter for all of them;
//   pass 2  camera sample and camera ray for each free slot that drew a work item inside the bounds (an item outside
//           them -- a pixel of an edge tile beyond the film -- leaves its slot idle for this iteration: the host stops
//           when no path is alive AND the work counter has passed the end); count new and continuing paths;
//   ------  one atomicAdd each on the two cursors of the extend work list;
//   pass 3  write the work list: new camera rays from the front (consecutive samples of a pixel: the most coherent rays),
//           continuing paths from the back.
__global__ void __launch_bounds__(BLOCK) k_generate(DScene s, Pool pool, float *film, DevCounters *ctr, WorkDesc wd) {
    __shared__ float sL[BLOCK * 33];
    __shared__ float sFilter[256];  // the 16x16 filter table, one LDS copy per block
    __shared__ unsigned sWant[SLOT_CHUNKS][BLOCK / 64], sPrim[SLOT_CHUNKS][BLOCK / 64], sCont[SLOT_CHUNKS][BLOCK / 64];
    __shared__ unsigned long long sWorkBase;
    __shared__ unsigned sPrimBase, sContBase, sTotWant, sTotFin;
    __shared__ unsigned sFinCnt[SLOT_CHUNKS][BLOCK / 64];
    __shared__ unsigned short sFin[SLOT_CHUNKS * BLOCK];   // the block's finished slots: offset in the block | 0x8000 if L reads as zero | 0x4000 if it lives in Q_LB
    static_assert(SLOT_CHUNKS * BLOCK <= 0x4000, "sFin: the offset shares its word with two flags");
    sFilter[threadIdx.x] = s.filterTable[threadIdx.x];
    __syncthreads();
    unsigned bad = 0, cam = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waveBase = threadIdx.x & ~63;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    const int nBands = s.nBands;
    unsigned wantBits = 0, restartBits = 0, contBits = 0, gotBits = 0;   // bit ch: state of this thread's slot in chunk ch
    unsigned finBits = 0, finZeroBits = 0, finInBBits = 0;
#ifdef MIPT_EXP_STAMPS
    unsigned long long stampLast = __builtin_amdgcn_s_memtime();
    const bool stampOn = (blockIdx.x % 61u) == 0u;
#endif
    // ---------------------------------------------------------------- pass 1: film flush
    // The finished paths are a third of the block's slots: they are listed first (in slot order: neighbours in the list are
    // samples of the same pixel more often than not) and flushed by all lanes together, as the refill below -- chunk by
    // chunk, a third of each wave worked through eight dependent round trips to memory (state word, then the L line) and
    // the kernel waited them out at four blocks per CU (in-kernel stamps, round 3: 78 % of the kernel in this pass).
    {
        int fl[SLOT_CHUNKS];
#pragma unroll
        for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {   // (all eight loads in flight together)
            const uint32_t slot = (blockIdx.x * SLOT_CHUNKS + ch) * BLOCK + threadIdx.x;
            fl[ch] = slot < pool.n ? pool.I(I_FLAGS, slot) : -1;
        }
#pragma unroll
        for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {
            const uint32_t slot = (blockIdx.x * SLOT_CHUNKS + ch) * BLOCK + threadIdx.x;
            const bool valid = slot < pool.n;
            const int flags = fl[ch];
            const bool fin = valid && (flags & F_FINISHED);
            // Integrator "spectralpath" (spectralpath.cpp:258-318): nBands paths per camera sample; a finished
            // path hands its bins to the sample's stitched spectrum and the slot restarts on the same camera ray
            // with the sampler dimension running on; the last band flushes the stitched spectrum.
            const bool restart = fin && nBands > 1 && pool.I(I_BAND, slot) + 1 < nBands;
            const bool want = valid && (fin || (flags & FLAG_MASK) == 0) && !restart;   // (a flushed slot is free)
            if (fin) finBits |= 1u << ch;
            if (fin && (flags & F_L_ZERO)) finZeroBits |= 1u << ch;
            if (fin && (flags & F_L_IN_B)) finInBBits |= 1u << ch;
            if (want) wantBits |= 1u << ch;
            if (restart) restartBits |= 1u << ch;
            if (valid && !fin && (flags & F_ALIVE)) contBits |= 1u << ch;
            const unsigned long long wm = __ballot(want), fm = __ballot(fin);
            if (lane == 0) { sWant[ch][wave] = (unsigned)__popcll(wm); sFinCnt[ch][wave] = (unsigned)__popcll(fm); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // exclusive prefix over (chunk, wave) and the block's range of the work list in one add
        unsigned tot = 0;
        for (int ch = 0; ch < SLOT_CHUNKS; ++ch)
            for (int w = 0; w < BLOCK / 64; ++w) { const unsigned n = sWant[ch][w]; sWant[ch][w] = tot; tot += n; }
        sWorkBase = tot ? atomicAdd(&ctr->nextWork, (unsigned long long)tot) : ~0ull;
        sTotWant = tot;
    }
    if (threadIdx.x == 64) {
        unsigned tot = 0;
        for (int ch = 0; ch < SLOT_CHUNKS; ++ch)
            for (int w = 0; w < BLOCK / 64; ++w) { const unsigned n = sFinCnt[ch][w]; sFinCnt[ch][w] = tot; tot += n; }
        sTotFin = tot;
    }
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {
        const bool fin = (finBits >> ch) & 1u;
        const unsigned long long fm = __ballot(fin);
        if (fin) sFin[sFinCnt[ch][wave] + (unsigned)__popcll(fm & ltMask)] =
            (unsigned short)((ch * BLOCK + threadIdx.x) | (((finZeroBits >> ch) & 1u) ? 0x8000u : 0u) | (((finInBBits >> ch) & 1u) ? 0x4000u : 0u));
    }
    __syncthreads();
    const unsigned totFin = sTotFin;
    STAMP(19);
#pragma unroll 1
    for (unsigned round = 0; round < (totFin + BLOCK - 1) / BLOCK; ++round) {
        const unsigned fi = round * BLOCK + threadIdx.x;
        const bool fin = fi < totFin;
        const unsigned fe = fin ? sFin[fi] : 0u;
        const uint32_t slot = blockIdx.x * SLOT_CHUNKS * BLOCK + (fe & 0x3fffu);
        float myFx = 0, myFy = 0;
        int myZero = 0;
        bool restart = false;
        if (fin) {
            int band = 0;
            if (nBands > 1) band = pool.I(I_BAND, slot);
            myFx = pool.F(P_FILMX, slot);   // (asked for with the L line: behind the guards, the film rows waited a round trip of their own)
            myFy = pool.F(P_FILMY, slot);
            // guards of SamplerIntegrator::Render, integrator.cpp:295-316
            float yy = 0.f;
            bool hasNaN = false;
            float *row = &sL[threadIdx.x * 33];
            const bool lZero = (fe & 0x8000u) != 0;
            const int lPlane = (fe & 0x4000u) ? Q_LB : Q_L;
            for (int c = 0; c < NQ; ++c) {
                float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!lZero) v4 = pool.Q(lPlane + c, slot);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int b = 4 * c + k;
                    if (b < MI_NSPEC) {
                        const float v = Get4(v4, k);
                        hasNaN |= isnanf_(v);
                        yy += s.cieY[b] * v;
                        row[b] = v;
                    }
                }
            }
            float y = YScale(yy);
            bool zero = false;
            if (hasNaN) zero = true;
            else if ((double)y < -1e-5) zero = true;
            else if (isinff(y)) zero = true;
            if (zero) ++bad;
            if (nBands > 1) {
                const int lo = s.bandDelta * band, hi = min(s.bandDelta * (band + 1), MI_NSPEC);
                for (int c = 0; c < NQ; ++c) {   // bins [lo, hi) of the stitched spectrum <- this band's path
                    if (4 * c + 3 < lo || 4 * c >= hi) continue;
                    float4 v4 = pool.Q(Q_LCA + c, slot);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int b = 4 * c + k;
                        if (b >= lo && b < hi) Set4(v4, k, zero ? 0.f : row[b]);
                    }
                    pool.Q(Q_LCA + c, slot) = v4;
                }
                zero = false;
                if (band + 1 < nBands) restart = true;
                else {
                    yy = 0.f;
                    for (int c = 0; c < NQ; ++c) {
                        const float4 v4 = pool.Q(Q_LCA + c, slot);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int b = 4 * c + k;
                            if (b < MI_NSPEC) {
                                const float v = Get4(v4, k);
                                yy += s.cieY[b] * v;
                                row[b] = v;
                            }
                        }
                    }
                    y = YScale(yy);
                }
            }
            if (!zero && !restart && y > s.maxSampleLuminance) {  // FilmTile::AddSample clamp, film.h:126-127
                const float scaleL = s.maxSampleLuminance / y;
                for (int b = 0; b < MI_NSPEC; ++b) row[b] *= scaleL;
            }
            myZero = zero ? 1 : 0;
        }
        // (a wave reads only the rows its own lanes staged, and its LDS accesses complete in order: nothing to wait for, the
        // fences keep the compiler from moving the accesses -- at workgroup scope each one also waited for every film atomic
        // in flight)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        STAMP(20);
        {
            const int half = lane >> 5, bin = lane & 31;
            const int filterTableSize = 16;
            const float invRx = 1 / s.filterRadius[0], invRy = 1 / s.filterRadius[1];
            const int w = s.croppedBounds[2] - s.croppedBounds[0];
            // The sample's footprint (FilmTile::AddSample, film.h:131-141). With a radius of at most half a pixel -- the box
            // filter of the BASELINE scenes -- it is one pixel for every sample that does not sit exactly on a pixel border.
            int p0x = 0, p0y = 0, p1x = 0, p1y = 0, myTarget = -1;
            float myFw = 0.f;
            if (fin && !restart) {
                const float dx = myFx - 0.5f, dy = myFy - 0.5f;
                p0x = (int)ceilf(dx - s.filterRadius[0]); p0y = (int)ceilf(dy - s.filterRadius[1]);
                p1x = (int)floorf(dx + s.filterRadius[0]) + 1; p1y = (int)floorf(dy + s.filterRadius[1]) + 1;
                p0x = max(p0x, s.croppedBounds[0]); p0y = max(p0y, s.croppedBounds[1]);
                p1x = min(p1x, s.croppedBounds[2]); p1y = min(p1y, s.croppedBounds[3]);
                if (p1x - p0x == 1 && p1y - p0y == 1) {
                    const float fy = absf((p0y - dy) * invRy * filterTableSize), fx = absf((p0x - dx) * invRx * filterTableSize);
                    const int iy = min((int)floorf(fy), filterTableSize - 1), ix = min((int)floorf(fx), filterTableSize - 1);
                    myFw = sFilter[iy * filterTableSize + ix];
                    myTarget = (p0x - s.croppedBounds[0]) + (p0y - s.croppedBounds[1]) * w;
                }
            }
            // Single-pixel samples: summed per pixel in LDS, one row update per pixel. Two pixels per pass, half-wave h
            // summing the rows of group h bin by bin.
            unsigned long long todo = __ballot(myTarget >= 0);
            while (todo) {
                const int j0 = __ffsll((long long)todo) - 1;
                const int t0 = __shfl(myTarget, j0, 64);
                const unsigned long long g0 = __ballot(myTarget == t0) & todo;
                todo &= ~g0;
                unsigned long long g1 = 0ull;
                int t1 = -1;
                if (todo) {
                    const int j1 = __ffsll((long long)todo) - 1;
                    t1 = __shfl(myTarget, j1, 64);
                    g1 = __ballot(myTarget == t1) & todo;
                    todo &= ~g1;
                }
                unsigned long long g = half ? g1 : g0;
                const int tgt = half ? t1 : t0;
                float acc = 0.f;
                const int nWalk = max(__popcll(g0), __popcll(g1));
                for (int it = 0; it < nWalk; ++it) {   // both halves walk their own group, in step (the shuffles need every lane)
                    const bool have = g != 0ull;
                    const int p = have ? __ffsll((long long)g) - 1 : 0;
                    g &= g - 1;
                    const float fw = __shfl(myFw, p, 64);
                    const int zero = __shfl(myZero, p, 64);
                    if (have) {
                        if (bin == 31) acc += fw;                                                // filterWeightSum += fw
                        else if (!zero) acc += (sL[(waveBase + p) * 33 + bin] * 1.f) * fw;      // contribSum += L * sampleWeight * fw
                    }
                }
#ifdef MIPT_EXP_NOFILMATOMIC
                if (tgt >= 0 && acc == 12345.678f) atomicAdd(film + (size_t)tgt * 32 + bin, acc);   // (timing experiment)
#else
                if (tgt >= 0 && acc != 0.f) atomicAdd(film + (size_t)tgt * 32 + bin, acc);   // (x + 0 == x: a black bin is not sent)
#endif
            }
            // Wider footprints (other filters, samples on a pixel border): one row update per sample and pixel reached.
            unsigned long long mask = __ballot(fin && !restart && myTarget < 0);
            while (mask) {
                const int j0 = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                int j1 = -1;
                if (mask) { j1 = __ffsll((long long)mask) - 1; mask &= mask - 1; }
                const int j = half ? j1 : j0;
                const int src = j >= 0 ? j : 0;
                const float pfx = __shfl(myFx, src, 64), pfy = __shfl(myFy, src, 64);
                const int zero = __shfl(myZero, src, 64);
                const int q0x = __shfl(p0x, src, 64), q0y = __shfl(p0y, src, 64), q1x = __shfl(p1x, src, 64), q1y = __shfl(p1y, src, 64);
                if (j >= 0) {
                    const float val = (bin < MI_NSPEC) ? sL[(waveBase + j) * 33 + bin] : 0.f;
                    const float dx = pfx - 0.5f, dy = pfy - 0.5f;
                    for (int y2 = q0y; y2 < q1y; ++y2) {
                        float fy = absf((y2 - dy) * invRy * filterTableSize);
                        int iy = min((int)floorf(fy), filterTableSize - 1);
                        for (int x2 = q0x; x2 < q1x; ++x2) {
                            float fx = absf((x2 - dx) * invRx * filterTableSize);
                            int ix = min((int)floorf(fx), filterTableSize - 1);
                            float fw = sFilter[iy * filterTableSize + ix];
                            size_t pix = (size_t)(x2 - s.croppedBounds[0]) + (size_t)(y2 - s.croppedBounds[1]) * w;
                            float *dst = film + pix * 32 + bin;
                            if (bin == 31) atomicAdd(dst, fw);                     // filterWeightSum += fw
                            else if (!zero && val != 0.f) atomicAdd(dst, (val * 1.f) * fw);   // contribSum += L * sampleWeight * fw (x + 0 == x: a black bin is not sent)
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the rows are read before the next round overwrites them
        __builtin_amdgcn_wave_barrier();
        STAMP(21);
    }
    __syncthreads();
    // ---------------------------------------------------------------- pass 2: refill
    // The free slots are a third of the block's slots, spread over all chunks: they are listed (in slot order, so work item
    // base + i still goes to the i-th free slot) and refilled by all lanes together -- the camera sample (index of the
    // Halton / Sobol' sample, five dimensions, the camera ray) runs on full waves.
    unsigned short *sFree = (unsigned short *)sL;                 // (pass 1's rows are dead now)
    unsigned char *sGot = (unsigned char *)(sL + SLOT_CHUNKS * BLOCK / 2);
#pragma unroll 1
    for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {
        const bool want = (wantBits >> ch) & 1u;
        const unsigned long long wm = __ballot(want);
        if (want) sFree[sWant[ch][wave] + (unsigned)__popcll(wm & ltMask)] = (unsigned short)(ch * BLOCK + threadIdx.x);
        sGot[ch * BLOCK + threadIdx.x] = 0;
    }
    __syncthreads();
    const unsigned totWant = sTotWant;
    const unsigned nRounds = (nBands > 1) ? SLOT_CHUNKS : 0;   // spectralpath: the restarting slots, chunk by chunk as before
    bool anyAlive = false;
#pragma unroll 1
    for (unsigned it = 0; it < nRounds + (totWant + BLOCK - 1) / BLOCK; ++it) {
        const bool restartRound = it < nRounds;
        unsigned e = 0;
        bool want = false, restart = false;
        if (restartRound) { e = it * BLOCK + threadIdx.x; restart = (restartBits >> it) & 1u; }
        else { const unsigned i = (it - nRounds) * BLOCK + threadIdx.x; want = i < totWant; if (want) e = sFree[i]; }
        const uint32_t slot = blockIdx.x * SLOT_CHUNKS * BLOCK + e;
        bool got = false;
        int px = 0, py = 0, band = 0;
        long long sampleNum = 0;
        int dimBefore = 0;
        if (restart) {  // next band of the same camera sample
            dimBefore = StateDim(pool.I(I_FLAGS, slot));   // (the band's path goes on with the sampler dimension the last one reached)
            const int pix = pool.I(I_PIXEL, slot);
            px = (int)(short)(pix & 0xffff); py = pix >> 16;
            sampleNum = pool.I(I_SAMPLE, slot);
            band = pool.I(I_BAND, slot);
            got = true;
        }
        if (want) {
            const unsigned long long w = sWorkBase + (unsigned long long)((it - nRounds) * BLOCK + threadIdx.x);
            if (w < wd.totalWork) {
                // work order: runs of wd.run consecutive samples of a pixel, pixel by pixel through the shard's tiles, then
                // the next run (so the lanes of a wave hold neighbouring samples of a few pixels: the most coherent camera
                // rays, and finished samples that can be summed before they reach the film)
                const unsigned run = (unsigned)wd.run;
                const unsigned long long perChunk = (unsigned long long)wd.nTilesShard * 256ull * run;
                const unsigned long long chunk = w / perChunk;
                const unsigned long long inChunk = w % perChunk;
                sampleNum = wd.sampleBegin + (long long)(chunk * run + (inChunk % run));
                unsigned rem = (unsigned)(inChunk / run);
                int tileLocal = rem >> 8, pix = rem & 255;
                int tile = wd.shardIndex + tileLocal * wd.shardCount;
                int tx = tile % wd.nTilesX, ty = tile / wd.nTilesX;
                int blk = pix >> 6, within = pix & 63;
                px = s.sampleBounds[0] + tx * 16 + (within & 7) + (blk & 1) * 8;
                py = s.sampleBounds[1] + ty * 16 + (within >> 3) + (blk >> 1) * 8;
                if (px < s.sampleBounds[2] && py < s.sampleBounds[3] && px >= s.pixelBounds[0] && px < s.pixelBounds[2] &&
                    py >= s.pixelBounds[1] && py < s.pixelBounds[3])
                    got = true;
            }
        }
        if (got) {
            // GetCameraSample (sampler.cpp:46-52): pFilm = dims 0,1; time = dim 2; pLens = dims 3,4
            float u0, u1, lu, lv;
            int dimAfter;
            const uint64_t index = CameraSampleDims(s, px, py, sampleNum, &u0, &u1, &lu, &lv, &dimAfter);
            float pfx = (float)px + u0, pfy = (float)py + u1;
            Ray ray;
            CameraRay(s, pfx, pfy, lu, lv, &ray);
            ++cam;
            pool.R(R_RAY0, slot) = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.tMax);
            pool.R(R_RAY1, slot) = make_float4(ray.d.x, ray.d.y, ray.d.z, 1.f);   // etaScale = 1
            pool.F(P_FILMX, slot) = pfx; pool.F(P_FILMY, slot) = pfy;

            if (s.storePixelSample) {
                pool.I(I_PIXEL, slot) = (px & 0xffff) | (py << 16);
                pool.I(I_SAMPLE, slot) = (int)sampleNum;
            }
            if (!(restart && s.samplerType >= MI_SAMPLER_RANDOM)) {   // (a restarted band draws on from where its stream stands)
                pool.I(I_IDXLO, slot) = (int)(uint32_t)index;
                if (!s.index32) pool.I(I_IDXHI, slot) = (int)(uint32_t)(index >> 32);
            }
            const bool pixelSampler = IsPixelSampler(s);
            if (!restart && pixelSampler) pool.I(I_DIM, slot) = dimAfter;
            if (nBands > 1) {
                pool.I(I_BAND, slot) = restart ? band + 1 : 0;
                if (!restart) for (int c = 0; c < NQ; ++c) pool.Q(Q_LCA + c, slot) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            pool.I(I_FLAGS, slot) = StateWord(F_ALIVE | F_L_ZERO | F_BETA_ONE | F_DIFF, 0, pixelSampler ? 0 : (restart ? dimBefore : dimAfter));   // L = 0, beta = 1, not stored
            sGot[e] = 1;
        } else if (want) pool.I(I_FLAGS, slot) = 0;   // stays free (its finished path has been flushed)
    }
    STAMP(22);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {
        const bool got = sGot[ch * BLOCK + threadIdx.x] != 0;
        if (got) gotBits |= 1u << ch;
        const bool isCont = ((contBits >> ch) & 1u) != 0;
        anyAlive |= got || isCont;
        // the extend work list: new camera rays first (lanes of a traversal wave then hold neighbouring samples)
        const unsigned long long pm = __ballot(got), cm = __ballot(isCont);
        if (lane == 0) { sPrim[ch][wave] = (unsigned)__popcll(pm); sCont[ch][wave] = (unsigned)__popcll(cm); }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned (*cnt)[BLOCK / 64] = threadIdx.x == 0 ? sPrim : sCont;
        unsigned tot = 0;
        for (int ch = 0; ch < SLOT_CHUNKS; ++ch)
            for (int w = 0; w < BLOCK / 64; ++w) { const unsigned n = cnt[ch][w]; cnt[ch][w] = tot; tot += n; }
        const unsigned base = tot ? atomicAdd(threadIdx.x == 0 ? &ctr->primCount.v : &ctr->contCount.v, tot) : 0;
        if (threadIdx.x == 0) sPrimBase = base; else sContBase = base;
    }
    if (__any(anyAlive) && lane == 0) ctr->alive.v = 1;   // (a flag, not a count: the host only asks whether any path is left)
    __syncthreads();
    // ---------------------------------------------------------------- pass 3: the extend work list
#pragma unroll 1
    for (int ch = 0; ch < SLOT_CHUNKS; ++ch) {
        const uint32_t slot = (blockIdx.x * SLOT_CHUNKS + ch) * BLOCK + threadIdx.x;
        const bool isPrim = (gotBits >> ch) & 1u, isCont = (contBits >> ch) & 1u;
        const unsigned long long pm = __ballot(isPrim), cm = __ballot(isCont);
        if (isPrim) pool.extQ[sPrimBase + sPrim[ch][wave] + (unsigned)__popcll(pm & ltMask)] = slot;
        if (isCont) pool.extQ[pool.n - 1 - (sContBase + sCont[ch][wave] + (unsigned)__popcll(cm & ltMask))] = slot;
    }
    STAMP(23);
    CountAdd(&Stats(ctr).cameraRays, cam);
    CountAdd(&Stats(ctr).badSamples, bad);
}

#endif   // MIPT_HAS_MAIN

// Transform::operator()(const SurfaceInteraction&), transform.cpp:262-297, with an instance's InstanceToWorld: the fields the
// path reads (SurfaceInteraction) and the ones textures and bump mapping read besides (TriShading).
DEV void InteractionToWorld(const mi_instance &in, SurfaceInteraction *si) {
    const float *m = in.i2w, *mInv = in.w2i;
    V3 pErr;
    si->p = XfPointErr2(m, si->p, si->pError, &pErr);
    si->pError = pErr;
    si->n = Normalize(XfNormal(mInv, si->n));
    si->wo = Normalize(XfVector(m, si->wo));
    si->dpdu = XfVector(m, si->dpdu);
    si->shN = Normalize(XfNormal(mInv, si->shN));
    si->shDpdu = XfVector(m, si->shDpdu);
    si->shN = Faceforward(si->shN, si->n);
}
DEV void ShadingToWorld(const mi_instance &in, TriShading *ts) {
    const float *m = in.i2w, *mInv = in.w2i;
    ts->dpdv = XfVector(m, ts->dpdv);
    ts->shDpdv = XfVector(m, ts->shDpdv);
    ts->dndu = XfNormal(mInv, ts->dndu);
    ts->dndv = XfNormal(mInv, ts->dndv);
}

// Build the SurfaceInteraction of a recorded hit.
DEV void HitInteraction(const DScene &s, int prim, const V3 &ro, const V3 &rd, float b0, float b1, float b2, SurfaceInteraction *si) {
    const mi_prim p = s.prims[prim];
#ifdef MIPT_EXP_FLATTRI
    if (p.shape >= 0) {   // (timing experiment: the interaction from the pre-gathered leaf record, no indexed N / UV gather)
        const float4 a = s.primTri[3 * prim], b = s.primTri[3 * prim + 1], c = s.primTri[3 * prim + 2];
        const V3 p0(a.x, a.y, a.z), p1(b.x, b.y, b.z), p2(c.x, c.y, c.z);
        si->p = b0 * p0 + b1 * p1 + b2 * p2;
        si->pError = gammaf(7) * V3(absf(b0 * p0.x) + absf(b1 * p1.x) + absf(b2 * p2.x), absf(b0 * p0.y) + absf(b1 * p1.y) + absf(b2 * p2.y), absf(b0 * p0.z) + absf(b1 * p1.z) + absf(b2 * p2.z));
        si->wo = Normalize(-rd);
        V3 n = Normalize(Cross(p0 - p2, p1 - p2)), du, dv;
        CoordinateSystem(n, &du, &dv);
        n = Faceforward(n, si->wo);
        si->n = n; si->shN = n; si->dpdu = du; si->shDpdu = du;
        return;
    }
#endif
    if (p.shape >= 0) TriInteraction(s, p.shape, b0, b1, b2, rd, si);
    else { float t; SphereInteraction(s.spheres[~p.shape], ro, rd, kInfinity, si, &t); }
}

// (the shading kernel's dimensions start after the camera sample's, so dim >= 5 here)

// ------------------------------------------------------------------ shade
// One path vertex per lane, for the slots of one material class (queue built by
// k_extend: NL = 2 for materials with <= 2 lobes, NL = 8 otherwise -- "sorted" shading:
// a wave runs one class of BSDF code and the common class keeps its lobe lists in
// registers). Spectra are streamed bin by bin; each of the three spectral passes
// (light sample, MIS sample, continuation) computes its values and its black/non-black
// decision in one sweep.
// The path's throughput quad c (see F_BETA_ONE).
DEV float4 LoadBeta(const Pool &pool, int c, uint32_t slot, bool betaOne) {
    float4 bt = make_float4(1.f, 1.f, 1.f, 1.f);
    if (!betaOne) bt = pool.Q(Q_BETA + c, slot);
    return bt;
}

// Whole-line stores of a spectrum from a shading wave. The memory side takes a store instruction as it comes: 64 lanes
// each storing 16 B of its own path's line leave L2 as 64 partial 64-B writes, and a spectrum written quad by quad
// costs 8 x 64 B of write traffic for 128 B of data (PMC: k_shade wrote 1.66 KB per vertex for 0.36 KB of state). So
// the lanes park their quads in an LDS tile and the wave stores them eight lanes per path, every store instruction
// covering whole 128-B lines. Works from divergent code: only the lanes that are active at the call take part, the
// k-th group of eight active lanes storing the lines of the k-th, (k+G)-th, ... path that has something to store.
struct SpectrumTile {
    float4 q[NQ][BLOCK];
    int slotOf[BLOCK];
    unsigned char laneOf[BLOCK], specOf[BLOCK];
};
// (`spectrum` may differ from lane to lane: it travels with the path)
DEV void StoreSpectrumLines(SpectrumTile &t, const Pool &pool, int spectrum, uint32_t slot, bool wrote) {
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long active = __ballot(1), wmask = __ballot(wrote);
    const int nW = __popcll(wmask), nGroups = __popcll(active) >> 3;
    if (nGroups == 0) {   // fewer than eight lanes here: each stores its own quads
        if (wrote) for (int c = 0; c < NQ; ++c) pool.Q(spectrum + c, slot) = t.q[c][threadIdx.x];
        return;
    }
    if (wrote) {
        const int rw = __popcll(wmask & lt);
        t.slotOf[wbase + rw] = (int)slot;
        t.laneOf[wbase + rw] = (unsigned char)lane;
        t.specOf[wbase + rw] = (unsigned char)spectrum;
    }
    // (the exchange stays inside the wave -- rows wbase .. wbase + 63 -- and a wave's LDS accesses complete in order: the
    // fences only keep the compiler from moving them. At workgroup scope each one was also a wait for every load and store
    // the wave had in flight, the second one for the line stores just issued)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile and the list are written before they are read
    __builtin_amdgcn_wave_barrier();
    const int ra = __popcll(active & lt), g = ra >> 3, c = ra & 7;
    if (g < nGroups)
        for (int e = g; e < nW; e += nGroups)
            pool.Q((int)t.specOf[wbase + e] + c, (uint32_t)t.slotOf[wbase + e]) = t.q[c][wbase + t.laneOf[wbase + e]];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // ... and read before the tile is reused
    __builtin_amdgcn_wave_barrier();
}

#ifndef MIPT_SHADE_WAVES_PER_EU
#define MIPT_SHADE_WAVES_PER_EU 4
#endif
// Timing experiments (tools/shade_experiments.sh; the films of these builds are wrong, only k_shade's time is read):
// MIPT_EXP_NOSPEC evaluates one quad of every spectral pass instead of eight, MIPT_EXP_NOSTORE leaves the spectra unstored.
#ifdef MIPT_EXP_NOSPEC
constexpr int EXP_NQ = 1;
#else
constexpr int EXP_NQ = NQ;
#endif
#ifdef MIPT_EXP_NOSTORE
#define EXP_STORE(x) false
#else
#define EXP_STORE(x) (x)
#endif

template <int NL, unsigned TM>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(MIPT_SHADE_WAVES_PER_EU, 8))) k_shade(DScene s, Pool pool, DevCounters *ctr, unsigned classes) {
    // the grid covers the queues of `classes` back to back, each padded to whole blocks
    unsigned blk = blockIdx.x, count = 0;
    int cls = -1;
    for (int c = 0; c < MAX_CLASSES; ++c) {
        if (!((classes >> c) & 1)) continue;
        const unsigned n = ctr->shadeCount[c].v, nb = (n + BLOCK - 1) / BLOCK;
        if (blk < nb) { cls = c; count = n; break; }
        blk -= nb;
    }
    const uint32_t qi = blk * BLOCK + threadIdx.x;
    constexpr bool HALTON_ONLY = (TM & TM_SAMPLERS) == 0;
#ifdef MIPT_NO_SIMPLE_SPECTRA
    constexpr bool SIMPLE_SPECTRA = false;
#else
    constexpr bool SIMPLE_SPECTRA = NL <= 2 && TM_SIMPLE_KINDS(TM);   // the straight-line spectral passes (d_bsdf.h, SimpleLobes)
#endif
    // more than two lobes: f is summed lobe by lobe into the lane's column of the spectrum tile (AccumulateF, d_bsdf.h)
#ifdef MIPT_NO_ACCUMULATE
    constexpr bool ACCUM = false;
#else
    constexpr bool ACCUM = NL > 2 || (TM & TM_TEXTURED) != 0;   // (image-textured lobes too: their spectra quad by quad, TexturedQuad)
#endif
#ifdef MIPT_FUSED_HALTON
    constexpr bool FUSED_HALTON = HALTON_ONLY;
#else
    constexpr bool FUSED_HALTON = false;   // (measured: the side-by-side digit loops cost 29 more scratch instructions and 7 % of the kernel)
#endif
    __shared__ SpectrumTile tile;
    auto rdTile = [&](int c) -> float4 { return tile.q[c][threadIdx.x]; };
    auto wrTile = [&](int c, const float4 &v) { tile.q[c][threadIdx.x] = v; };
#ifdef MIPT_EXP_STAMPS
    unsigned long long stampLast = __builtin_amdgcn_s_memtime();
    const bool stampOn = (blockIdx.x % 61u) == 0u;   // (one block in 61 reports: the atomics of every wave would be the kernel)
#endif
    unsigned totalPaths = 0, pathLen = 0, zeroNow = 0;
    bool wantShadow = false, wantMis = false;
    uint32_t slot = 0;
    if (cls >= 0 && qi < count) {
        slot = pool.shadeQ[(size_t)cls * pool.n + qi];
        const int word = pool.I(I_FLAGS, slot);   // flags | bounces | sampler dimension (StateWord)
        const int flags = word & FLAG_MASK;
        bool lZero = (flags & F_L_ZERO) != 0, betaWritten = false;
        const int lPlane = LPlane(flags);   // the line that holds the path's L
        const bool betaOne = (flags & F_BETA_ONE) != 0;
        const int bounces = StateBounces(word);
        int dimNow = StateDim(word);
        const int prim = pool.I(I_HITPRIM, slot);
        const bool found = prim >= 0;
        auto loadBeta = [&](int c) -> float4 { return LoadBeta(pool, c, slot, betaOne); };
        // the ray and the interaction are needed by vertices that will be shaded or may show emitted light; an escaped
        // ray without environment lights and a path at its last vertex need neither (a third of the queue entries)
        const bool emitCheck = bounces == 0 || (flags & F_SPECULAR);
        const bool needIsect = found && (bounces < s.maxDepth || emitCheck);
        const bool needRay = needIsect || (!found && emitCheck && TM_LIGHT(TM, MI_LIGHT_INFINITE) && s.nInfiniteLights > 0);
        STAMP(1);
        float4 ray0 = make_float4(0.f, 0.f, 0.f, 0.f), ray1 = make_float4(0.f, 0.f, 1.f, 1.f);
        if (needRay) { ray0 = pool.R(R_RAY0, slot); ray1 = pool.R(R_RAY1, slot); }
        V3 ro(ray0.x, ray0.y, ray0.z), rd(ray1.x, ray1.y, ray1.z);
        SurfaceInteraction isect;
        bool finished = false, passThrough = false;
        // a hit inside an object instance: the interaction is built with the ray in the instance's space and taken to the
        // world by InstanceToWorld (TransformedPrimitive::Intersect, primitive.cpp:78-92)
        int inst = -1;
        V3 roS = ro, rdS = rd;   // the ray in the space the hit shape was intersected in
        if constexpr ((TM & TM_INSTANCES) != 0) {
            if (needIsect) inst = pool.I(I_HITINST, slot);
            if (inst >= 0) { const Ray ir = XfRay(s.instances[inst].w2i, Ray(ro, rd, kInfinity)); roS = ir.o; rdS = ir.d; }
        }
        if (needIsect) { const float4 hr = pool.R(R_HIT, slot); HitInteraction(s, prim, roS, rdS, hr.y, hr.z, hr.w, &isect); }
        STAMP(2);
        SurfaceInteraction isectObj;
        if constexpr ((TM & TM_INSTANCES) != 0) {
            if (inst >= 0) { isectObj = isect; InteractionToWorld(s.instances[inst], &isect); }
        }
        // emitted light at the vertex, path.cpp:91-101
        if ((bounces == 0 || (flags & F_SPECULAR)) && found) {
            const int li = s.prims[prim].area_light;
            if (li >= 0) {
                const mi_light &l = s.lights[li];
                if (l.two_sided || Dot(isect.n, -rd) > 0)
{
#pragma unroll 1
                    for (int c = 0; c < NQ; ++c) {
                        const float4 bt = loadBeta(c);
                        float4 L4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (!lZero) L4 = pool.Q(lPlane + c, slot);
                        const float4 Le = LoadSpec4(l.L, c);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int b = 4 * c + k;
                            if (b < MI_NSPEC) Set4(L4, k, Get4(L4, k) + Get4(bt, k) * Get4(Le, k));
                        }
                        pool.Q(lPlane + c, slot) = L4;
                    }
                    lZero = false;
                }
            }
        }
        if ((bounces == 0 || (flags & F_SPECULAR)) && !found) {  // escaped: scene.infiniteLights, path.cpp:96-99
            for (int il = 0; TM_LIGHT(TM, MI_LIGHT_INFINITE) && il < s.nInfiniteLights; ++il) {
                const IllumRGB le = InfiniteLe(s, s.lights[s.infiniteLights[il]], rd);
#pragma unroll 1
                for (int c = 0; c < NQ; ++c) {
                    const float4 bt = loadBeta(c);
                    float4 L4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!lZero) L4 = pool.Q(lPlane + c, slot);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int b = 4 * c + k;
                        if (b < MI_NSPEC) Set4(L4, k, Get4(L4, k) + Get4(bt, k) * IllumBin(s, le, b));
                    }
                    pool.Q(lPlane + c, slot) = L4;
                }
                lZero = false;
            }
        }
        if (!found || bounces >= s.maxDepth) finished = true;
        STAMP(3);
        int newFlags = 0;
        if (!finished && s.prims[prim].material < 0) {  // interface without BSDF: continue through it, path.cpp:108-113
            Ray r = SpawnRay(isect, rd);
            pool.R(R_RAY0, slot) = make_float4(r.o.x, r.o.y, r.o.z, r.tMax);
            passThrough = true;  // flags and bounce count stay as they are (but the spawned ray has no differentials)
            if (flags & F_DIFF) pool.I(I_FLAGS, slot) = word & ~F_DIFF;
        }
        if (!finished && !passThrough) {
            const mi_material *mat = &s.materials[s.prims[prim].material];
            BSDFFrame fr;
            fr.m = mat;
            fr.mask = 0xffu;
            fr.ov.u = fr.ov.v = 0.f; fr.ov.onU = fr.ov.onV = false;
            fr.ov.sigMode = 0; fr.ov.sigA = fr.ov.sigB = 0.f;
            fr.ov.disney = false; fr.ov.rough = 0.f;
            LobeTexT<NL> lt;
            const LobeTexT<NL> *ltp = nullptr;
            if constexpr ((TM & TM_TEXTURED) != 0) {
                // Material::ComputeScatteringFunctions with image textures: evaluate them at the hit, keep the lobes
                // whose tested spectrum is not black (matte.cpp:55-63, plastic.cpp:52-68, uber.cpp:60-100, ...)
                lt.hasR = lt.hasS = lt.mulR = lt.mulS = 0u;
                lt.rules = 0u; lt.lum = 0.f; lt.hasK = 0u;
                lt.basis = s.rgbIllum; lt.textures = s.textures;
                ltp = &lt;
                float u = 0.f, v = 0.f;
                TriShading tsh;
                bool haveUV = false;
                if (mat->textured) {
                    const int shape = s.prims[prim].shape;
                    if (shape >= 0) {
                        const float4 hr = pool.R(R_HIT, slot);
                        TriTexCoords(s, shape, hr.y, hr.z, hr.w, inst >= 0 ? isectObj : isect, &u, &v, &tsh);
                        haveUV = true;
                    } else
                        haveUV = SphereTexCoords(s.spheres[~shape], roS, rdS, &u, &v, &tsh);
                    if constexpr ((TM & TM_INSTANCES) != 0) {
                        if (inst >= 0) ShadingToWorld(s.instances[inst], &tsh);
                    }
                }
                if (haveUV) {
                    TexDifferentials td;
                    td.dudx = td.dvdx = td.dudy = td.dvdy = 0;
                    if (flags & F_DIFF) {   // SurfaceInteraction::ComputeDifferentials, interaction.cpp:99-143
                        float lu = 0.f, lv = 0.f;
                        if (s.camera.lens_radius > 0) {   // the camera sample's lens position again
                            const int pixw = pool.I(I_PIXEL, slot);
                            float cu0, cu1;
                            CameraSampleDims(s, (int)(short)(pixw & 0xffff), pixw >> 16, (long long)pool.I(I_SAMPLE, slot), &cu0, &cu1, &lu, &lv);
                        }
                        const CamDifferentials cd = CameraDifferentials(s, pool.F(P_FILMX, slot), pool.F(P_FILMY, slot), lu, lv, ro, rd, s.invSqrtSpp);
                        td = ComputeDifferentials(isect.p, isect.n, isect.dpdu, tsh.dpdv, cd);
                    }
                    if (mat->bump_tex >= 0) Bump(s, mat->bump_tex, u, v, td, tsh, &isect);   // `if (bumpMap) Bump(bumpMap, si)`
                    // `rough = roughness->Evaluate(*si); if (remapRoughness) rough = RoughnessToAlpha(rough)` (plastic.cpp:57-62 ...)
                    float rawU = mat->bxdf[0].p[6], rawV = mat->bxdf[0].p[7];   // (MI_ROUGH_GLASS: the values before the remap)
                    if (mat->rough_flags & MI_ROUGH_DISNEY) {   // `Float rough = roughness->Evaluate(*si)`, disney.cpp:491
                        fr.ov.disney = true;
                        fr.ov.rough = EvalFloatImageTexture(s, mat->rough_tex[0], u, v, td);
                    } else
                    if (mat->rough_tex[0] >= 0) {
                        const float rv = EvalFloatImageTexture(s, mat->rough_tex[0], u, v, td);
                        fr.ov.u = (mat->rough_flags & MI_ROUGH_REMAP) ? RoughnessToAlpha(rv) : rv;
                        fr.ov.onU = true;
                        rawU = rv;
                    }
                    if (mat->rough_tex[1] >= 0) {
                        if (mat->rough_tex[1] == mat->rough_tex[0]) { fr.ov.v = fr.ov.u; rawV = rawU; }
                        else {
                            const float rv = EvalFloatImageTexture(s, mat->rough_tex[1], u, v, td);
                            fr.ov.v = (mat->rough_flags & MI_ROUGH_REMAP) ? RoughnessToAlpha(rv) : rv;
                            rawV = rv;
                        }
                        fr.ov.onV = true;
                    }
                    if (mat->sigma_tex >= 0) {   // `Float sig = Clamp(sigma->Evaluate(*si), 0, 90)`, matte.cpp:57; OrenNayar's constructor
                        const float sig = clampf(EvalFloatImageTexture(s, mat->sigma_tex, u, v, td), 0.f, 90.f);
                        if (sig == 0) fr.ov.sigMode = 2;
                        else {
                            const float sigma = (kPi / 180) * sig, sigma2 = sigma * sigma;
                            fr.ov.sigMode = 1;
                            fr.ov.sigA = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                            fr.ov.sigB = 0.45f * sigma2 / (sigma2 + 0.09f);
                        }
                    }
                    unsigned mask = 0u;
                    for (int i = 0; i < mat->n_bxdfs; ++i) {
                        const mi_lobe_tex ltx = mat->tex[i];
                        if (ltx.tex_R < 0 && ltx.tex_S < 0) { mask |= 1u << i; continue; }
                        if (i >= NL) continue;
                        if (ltx.tex_R >= 0) {
                            lt.r[i] = EvalImageTexture(s, ltx.tex_R, u, v, td);
                            if (ltx.rule == MI_LOBE_METAL) lt.hasK |= 1u << i;   // (k's texture: the lobe's R stays the constant)
                            else lt.hasR |= 1u << i;
                            if (ltx.flags & MI_LOBE_TEX_MUL_R) lt.mulR |= 1u << i;
                        }
                        if (ltx.tex_S >= 0) {
                            lt.s[i] = EvalImageTexture(s, ltx.tex_S, u, v, td);
                            lt.hasS |= 1u << i;
                            if (ltx.flags & MI_LOBE_TEX_MUL_S) lt.mulS |= 1u << i;
                        }
                        lt.rules |= (unsigned)(ltx.rule & 15) << (4 * i);
                        if (ltx.rule >= MI_LOBE_ALWAYS) {   // "disney": lobes are added whatever the colour is
                            if (ltx.rule >= MI_LOBE_DISNEY_SHEEN && ltx.rule <= MI_LOBE_DISNEY_STRANS && lt.lum == 0.f) {   // lum = c.y(), once per vertex (every lobe has the same colour)
                                float yy = 0.f;
                                for (int b = 0; b < MI_NSPEC; ++b) yy += SpecYBinAccum(s, b, TexBin(lt.basis, lt.textures, lt.r[i], b));
                                lt.lum = YScale(yy);
                            }
                            mask |= 1u << i;
                            continue;
                        }
                        // (the spectrum the material tests with IsBlack(), a quad of bins at a time; only what the lobe's rule reads)
                        unsigned rNZ = 0u, sNZ = 0u, texNZ = 0u;
                        auto orQuad = [](unsigned &acc, const float4 &q, int c) {
                            OrNonZero(acc, q.x); OrNonZero(acc, q.y); OrNonZero(acc, q.z);
                            if (c != NQ - 1) OrNonZero(acc, q.w);   // bin 31 does not exist
                        };
#pragma unroll 1
                        for (int c = 0; c < NQ; ++c) {
                            if (ltx.rule != MI_LOBE_IF_TEX) orQuad(rNZ, TexturedQuad(lt, mat->bxdf[i], i, 0, c), c);
                            if (ltx.rule == MI_LOBE_IF_R_OR_S) orQuad(sNZ, TexturedQuad(lt, mat->bxdf[i], i, 1, c), c);
                            if (ltx.rule == MI_LOBE_IF_TEX) {
                                if (ltx.tex_R >= 0) orQuad(texNZ, TexQuad(lt.basis, lt.textures, lt.r[i], c), c);
                                if (ltx.tex_S >= 0) orQuad(texNZ, TexQuad(lt.basis, lt.textures, lt.s[i], c), c);
                            }
                        }
                        const bool rNonBlack = rNZ != 0u, sNonBlack = sNZ != 0u, texNonBlack = texNZ != 0u;
                        const bool present = ltx.rule == MI_LOBE_IF_R_OR_S ? (rNonBlack || sNonBlack) : (ltx.rule == MI_LOBE_IF_TEX ? texNonBlack : rNonBlack);
                        if (present) mask |= 1u << i;
                    }
                    if (mat->rough_flags & MI_ROUGH_GLASS) {   // glass.cpp:66: `isSpecular = urough == 0 && vrough == 0` at this hit
                        const bool spec = rawU == 0 && rawV == 0;
                        mask &= spec ? 1u : ~1u;   // lobe 0 is the FresnelSpecular one
                    }
                    fr.mask = mask;
                }
            }
            // BSDF ctor, reflection.h:170-176 (after Bump(): it reads the shading geometry)
            fr.ns = isect.shN; fr.ng = isect.n; fr.ss = Normalize(isect.shDpdu); fr.ts = Cross(fr.ns, fr.ss);
            PathSampler ps;
            ps.index = (uint32_t)pool.I(I_IDXLO, slot);
            if (!s.index32) ps.index |= (uint64_t)(uint32_t)pool.I(I_IDXHI, slot) << 32;
            ps.dim = dimNow;
            if constexpr (!HALTON_ONLY) { if (IsPixelSampler(s)) ps.dim = pool.I(I_DIM, slot); }
            const int *__restrict__ pixelPlane = pool.i + (size_t)I_PIXEL * pool.n, *__restrict__ samplePlane = pool.i + (size_t)I_SAMPLE * pool.n;
            const int nonSpec = MI_BSDF_ALL & ~MI_BSDF_SPECULAR;
        STAMP(4);
            // ---- direct lighting: UniformSampleOneLight + EstimateDirect, integrator.cpp:85-215
            if (NumComponents(fr, nonSpec) > 0) {
                ++totalPaths;
                newFlags |= F_NEE;
                if (s.nLights > 0) {
                    const uint32_t di = LightDistribIndex(s, isect.p);
                    float selPdf;
                    // the five dimensions of the estimate (light choice, light sample, BSDF sample): with the Halton sampler
                    // -- values are a pure function of (index, dimension) -- their digit loops run side by side
                    float u5[5];
                    if constexpr (FUSED_HALTON) { ScrambledDimensionsFused<5>(s.primes, s.primeSums, s.perms, s.primeMagic, ps.index, ps.dim, u5); ps.dim += 1; }
                    else u5[0] = Get1D<HALTON_ONLY>(s, ps, pixelPlane, samplePlane, slot);
                    const int lightNum = SampleDiscrete(s.ldFunc + (size_t)di * s.nLights, s.ldCdf + (size_t)di * (s.nLights + 1),
                                                        s.ldFuncInt[di], (int)s.nLights, u5[0], &selPdf);
                    if (selPdf != 0) {
                        if constexpr (FUSED_HALTON) ps.dim += 4;
                        else {   // uLight = Get2D(), uScattering = Get2D() (integrator.cpp:100-101)
                            Get2D<HALTON_ONLY>(s, ps, pixelPlane, samplePlane, slot, &u5[1], &u5[2]);
                            Get2D<HALTON_ONLY>(s, ps, pixelPlane, samplePlane, slot, &u5[3], &u5[4]);
                        }
                        const float uL0 = u5[1], uL1 = u5[2], uS0 = u5[3], uS1 = u5[4];
        STAMP(5);
                        const mi_light &light = s.lights[lightNum];
                        const bool selIsOne = (selPdf == 1.f);  // x / 1 == x: skip the division
                        const Divisor selDiv = MakeDivisor(selPdf);
                        const LightSample ls = SampleLi<TM>(s, light, isect, uL0, uL1);
        STAMP(6);
                        const float lightPdf = ls.pdf;
                        if (lightPdf > 0 && !ls.black) {
                            BSDFEvalT<NL> ev;
                            if constexpr (ACCUM) AccumulateF<NL, TM>(fr, isect.wo, ls.wi, nonSpec, ltp, rdTile, wrTile);
                            else BSDF_f<NL, TM>(fr, isect.wo, ls.wi, nonSpec, &ev);
                            auto fQuad = [&](int c) -> float4 { if constexpr (ACCUM) return rdTile(c); else return EvalQuad<NL, TM>(ev, mat->bxdf, c, ltp); };
                            const float absdot = AbsDot(ls.wi, isect.shN);
                            const float scatteringPdf = BSDF_Pdf<TM>(fr, isect.wo, ls.wi, nonSpec);
        STAMP(7);
                            const bool delta = IsDeltaLight(light);
                            float weight = 1.f;
                            if (!delta) { float pf = 1 * lightPdf, pg = 1 * scatteringPdf; weight = (pf * pf) / (pf * pf + pg * pg); }
                            const Divisor lpDiv = MakeDivisor(lightPdf);
                            bool fNonBlack = false, liNonBlack = false, nzAny = false;
                            // the path's L so far, quad c: the candidate is L + contribution (F_CAND)
                            auto lBase = [&](int c) -> float4 { return lZero ? make_float4(0.f, 0.f, 0.f, 0.f) : pool.Q(lPlane + c, slot); };
                            auto neeQuad = [&](int c, const float4 &fq, const float4 &Lq) {
                                const float4 bt = loadBeta(c), l4 = lBase(c);
                                float4 out = l4;
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const int b = 4 * c + k;
                                    if (b < MI_NSPEC) {
                                        const float f = Get4(fq, k) * absdot;
                                        const float Li = Get4(Lq, k);
                                        fNonBlack |= (f != 0.f);
                                        liNonBlack |= (Li != 0.f);
                                        float Ld = delta ? DivBy(f * Li, lpDiv) : DivBy((f * Li) * weight, lpDiv);
                                        if (!selIsOne) Ld = DivBy(Ld, selDiv);
                                        const float contrib = Get4(bt, k) * Ld;
                                        nzAny |= (contrib != 0.f);
                                        Set4(out, k, Get4(l4, k) + contrib);
                                    }
                                }
                                tile.q[c][threadIdx.x] = out;
                            };
                            if constexpr (SIMPLE_SPECTRA) {
                                // straight-line form (d_bsdf.h, SimpleLobes); a quad in which a quotient leaves DivBy's fast range is redone with DivBy
                                const SimpleLobes<NL> sl = MakeSimpleLobes<NL, TM>(ev);
                                const mi_bxdf *bx = mat->bxdf;
                                const bool allSelOne = __all(selIsOne);
                                const bool divFast = sl.divisorsFast && lpDiv.fast && (selIsOne || selDiv.fast);
                                unsigned accF = 0u, accLi = 0u, accNz = 0u;
                                auto quad = [&](int c, auto exactTag) {
                                    constexpr bool EXACT = decltype(exactTag)::value;
                                    DivTrack trk;
                                    trk.Reset(divFast);
                                    const float4 bt = loadBeta(c), l4 = lBase(c);
                                    float4 fq = SimpleEvalQuad<NL, TM, EXACT>(sl, bx, c, trk), Lq = LiQuad<TM>(s, light, ls, c);
                                    if (c == NQ - 1) { fq.w = 0.f; Lq.w = 0.f; }   // bin 31 does not exist
                                    unsigned aF = 0u, aLi = 0u, aNz = 0u;
                                    float o[4];
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        const float f = Get4(fq, k) * absdot;
                                        const float Li = Get4(Lq, k);
                                        OrNonZero(aF, f); OrNonZero(aLi, Li);
                                        const float x = (f * Li) * weight;   // (weight == 1 for a delta light: x * 1 == x)
                                        float Ld;
                                        if constexpr (EXACT) { Ld = DivBy(x, lpDiv); if (!selIsOne) Ld = DivBy(Ld, selDiv); }
                                        else {
                                            Ld = DivFast(x, lpDiv.d, lpDiv.r, trk);
                                            if (!allSelOne) { const float Ld2 = DivFast(Ld, selDiv.d, selDiv.r, trk); Ld = selIsOne ? Ld : Ld2; }
                                        }
                                        o[k] = Get4(bt, k) * Ld;
                                        OrNonZero(aNz, o[k]);
                                    }
                                    if (!EXACT && __any(trk.Bad())) return false;
                                    tile.q[c][threadIdx.x] = make_float4(l4.x + o[0], l4.y + o[1], l4.z + o[2], l4.w + o[3]);
                                    accF |= aF; accLi |= aLi; accNz |= aNz;
                                    return true;
                                };
#pragma unroll 1
                                for (int c = 0; c < EXP_NQ; ++c)
                                    if (!quad(c, std::false_type{})) quad(c, std::true_type{});
                                fNonBlack |= accF != 0u; liNonBlack |= accLi != 0u; nzAny |= accNz != 0u;
                            } else {
#pragma unroll 1
                                for (int c = 0; c < EXP_NQ; ++c) neeQuad(c, fQuad(c), LiQuad<TM>(s, light, ls, c));
                            }
                            // the tile holds L + contribution, the candidate (F_CAND): only one whose shadow ray will be traced
                            // is ever read
                            const bool traced = fNonBlack && liNonBlack;
        STAMP(8);
                            StoreSpectrumLines(tile, pool, LOtherPlane(flags), slot, EXP_STORE(traced));
                            if (traced) newFlags |= F_CAND | (nzAny ? F_NEE_NZ : 0);
                            if (traced) {  // the shadow ray is traced iff f != 0 (integrator.cpp:138-150)
                                Ray sr = SpawnRayTo(isect, ls.pLight);
                                pool.R(R_SH0, slot) = make_float4(sr.o.x, sr.o.y, sr.o.z, sr.d.x);
                                pool.R(R_SH1, slot) = make_float4(sr.d.y, sr.d.z, 0.f, 0.f);
                                newFlags |= F_SHADOW;
                            }
        STAMP(9);
                        }
                        if (!IsDeltaLight(light)) {  // BSDF sampling with MIS, integrator.cpp:167-213
                            V3 wi;
                            float sPdf = 0;
                            int sampledType = 0;
                            BSDFEvalT<NL> ev;
                            V3 wiLocal;
                            const bool ok = BSDF_Sample_f<NL, TM, !ACCUM>(fr, isect.wo, &wi, uS0, uS1, &sPdf, nonSpec, &sampledType, &ev, &wiLocal);
                            auto fQuad = [&](int c) -> float4 { if constexpr (ACCUM) return rdTile(c); else return EvalQuad<NL, TM>(ev, mat->bxdf, c, ltp); };
        STAMP(10);
                            if (ok && sPdf > 0) {
                                const float absdot = AbsDot(wi, isect.shN);
                                // Pdf_Li has no side effect: evaluate it before knowing whether f is black
                                float weight = 1;
                                bool go = true;
                                float tShape = -1.f;   // a triangle light: where wi meets it
                                if (!(sampledType & MI_BSDF_SPECULAR)) {
                                    const float lp = (TM_LIGHT(TM, MI_LIGHT_INFINITE) && light.type == MI_LIGHT_INFINITE) ? InfinitePdfLi(s, light, wi)
                                                                                       : ShapePdf(s, light.shape, light.area, isect, wi, &tShape);
                                    if (lp == 0) go = false;
                                    else { float pf = 1 * sPdf, pg = 1 * lp; weight = (pf * pf) / (pf * pf + pg * pg); }
                                }
                                const bool isEnvLight = TM_LIGHT(TM, MI_LIGHT_INFINITE) && light.type == MI_LIGHT_INFINITE;
                                IllumRGB envLe;
                                envLe.i1 = envLe.i2 = 0; envLe.w0 = envLe.w1 = envLe.w2 = 0;
                                if (isEnvLight && go) envLe = InfiniteLe(s, light, wi);   // light.Le(ray) when the ray escapes
                                const Divisor spDiv = MakeDivisor(sPdf);
                                bool fNonBlack = false;
                                // a ray that cannot reach the sampled area light is traced all the same (the reference
                                // traces it), but its contribution is never read: F_MIS_DARK
                                const Ray mr = SpawnRay(isect, wi);
                                float spanLo = 0.f, spanHi = kInfinity;
                                const bool dark = go && !isEnvLight && !RaySpanInBox(mr.o, mr.d, s.lightBounds[2 * lightNum], s.lightBounds[2 * lightNum + 1], &spanLo, &spanHi);
                                float misTHi = kInfinity, misWord = __uint_as_float(MIS_EXCL_NONE | (31u << MIS_EXCL_BITS));   // (environment light, dark ray: any hit ends it)
                                if (s.misAny && go && !dark && !isEnvLight) {
                                    // a triangle: Shape::Pdf has intersected it (the same test the traversal runs); a sphere: its span in the
                                    // dilated bounds. Either way widened by what the float arithmetic of the triangle and box tests can disagree
                                    // by about where things are along the ray: those work on coordinates relative to the ray's origin, so their
                                    // absolute error in t grows with the distance of the farthest vertex, a few ulps of it (a ray that starts
                                    // 1e-3 above a 2000-unit ground plane meets it at a t that is 10 % noise, and the plane's box later than that:
                                    // tests/test_gpu_parity.py::test_mis_visibility_kernel_verdicts_on_recorded_rays). 64 ulps of the distance to
                                    // the far corner of the world bound: inside the span the reference's visiting order decides, outside it cannot.
                                    const float far = maxf(absf(mr.o.x - s.wbMin[0]), absf(s.wbMax[0] - mr.o.x)) + maxf(absf(mr.o.y - s.wbMin[1]), absf(s.wbMax[1] - mr.o.y)) +
                                                      maxf(absf(mr.o.z - s.wbMin[2]), absf(s.wbMax[2] - mr.o.z));
#ifdef MIPT_EXP_NO_SPAN_SLACK   // (experiment build; no test scene so far tells it from the default)
                                    const float slack = 0.f * far;
#else
                                    const float slack = far * 0x1p-18f;
#endif
                                    if (tShape > 0.f) MisSpanWords(tShape * (1.f - 1e-5f) - slack, tShape * (1.f + 1e-5f) + slack, (unsigned)s.lightPrim[lightNum], &misTHi, &misWord);
                                    else MisSpanWords(spanLo * 0.9999f - slack, spanHi + slack, (unsigned)s.lightPrim[lightNum], &misTHi, &misWord);
                                }
                                if constexpr (ACCUM) {   // f of the sampled direction into the tile (only if something will read it)
                                    if (go) {
                                        if (ev.n > 0) AccumulateSpecular<NL, TM>(ev.lobes[0], mat->bxdf, ltp, rdTile, wrTile);
                                        else AccumulateFLocal<NL, TM>(fr, fr.WorldToLocal(isect.wo), wiLocal, Dot(wi, fr.ng) * Dot(isect.wo, fr.ng) > 0, nonSpec, ltp, rdTile, wrTile);
                                    }
                                }
        STAMP(11);
                                auto darkQuad = [&](int c, const float4 &fq) {   // (only: is f black? -- that decides whether the ray exists)
#pragma unroll
                                    for (int k = 0; k < 4; ++k)
                                        if (4 * c + k < MI_NSPEC) fNonBlack |= (Get4(fq, k) * absdot != 0.f);
                                };
                                // (when the light's pdf for wi is 0 the estimate ends here, integrator.cpp:186-187:
                                // nothing reads the spectrum then, so it is not formed)
                                auto misQuad = [&](int c, const float4 &fq, const float4 &Lq) {
                                    const float4 bt = loadBeta(c);
                                    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        const int b = 4 * c + k;
                                        if (b < MI_NSPEC) {
                                            const float f = Get4(fq, k) * absdot;
                                            fNonBlack |= (f != 0.f);
                                            const float LiB = isEnvLight ? IllumBin(s, envLe, b) : Get4(Lq, k);   // Le of the light if the ray reaches it
                                            float Ld = DivBy((f * LiB) * weight, spDiv);  // f * Li * Tr(=1) * weight / scatteringPdf
                                            if (!selIsOne) Ld = DivBy(Ld, selDiv);
                                            Set4(out, k, Get4(bt, k) * Ld);
                                        }
                                    }
                                    tile.q[c][threadIdx.x] = out;
                                };
                                const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
                                if constexpr (SIMPLE_SPECTRA) {
                                    const SimpleLobes<NL> sl = MakeSimpleLobes<NL, TM>(ev);
                                    const mi_bxdf *bx = mat->bxdf;
                                    const bool allSelOne = __all(selIsOne);
                                    const bool divFast = sl.divisorsFast && spDiv.fast && (selIsOne || selDiv.fast);
                                    unsigned accF = 0u;
                                    if (dark) {
                                        auto quad = [&](int c, auto exactTag) {
                                            constexpr bool EXACT = decltype(exactTag)::value;
                                            DivTrack trk;
                                            trk.Reset(sl.divisorsFast);
                                            float4 fq = SimpleEvalQuad<NL, TM, EXACT>(sl, bx, c, trk);
                                            if (c == NQ - 1) fq.w = 0.f;
                                            unsigned aF = 0u;
#pragma unroll
                                            for (int k = 0; k < 4; ++k) OrNonZero(aF, Get4(fq, k) * absdot);
                                            if (!EXACT && __any(trk.Bad())) return false;
                                            accF |= aF;
                                            return true;
                                        };
#pragma unroll 1
                                        for (int c = 0; c < EXP_NQ; ++c)
                                            if (!quad(c, std::false_type{})) quad(c, std::true_type{});
                                    }
                                    if (go && !dark) {
                                        auto quad = [&](int c, auto exactTag) {
                                            constexpr bool EXACT = decltype(exactTag)::value;
                                            DivTrack trk;
                                            trk.Reset(divFast);
                                            const float4 bt = loadBeta(c);
                                            float4 fq = SimpleEvalQuad<NL, TM, EXACT>(sl, bx, c, trk), Lq = isEnvLight ? zero4 : LoadSpec4(light.L, c);
                                            if (c == NQ - 1) { fq.w = 0.f; Lq.w = 0.f; }
                                            unsigned aF = 0u;
                                            float o[4];
#pragma unroll
                                            for (int k = 0; k < 4; ++k) {
                                                const float f = Get4(fq, k) * absdot;
                                                OrNonZero(aF, f);
                                                const float LiB = isEnvLight ? ((4 * c + k < MI_NSPEC) ? IllumBin(s, envLe, min(4 * c + k, MI_NSPEC - 1)) : 0.f) : Get4(Lq, k);
                                                const float x = (f * LiB) * weight;
                                                float Ld;
                                                if constexpr (EXACT) { Ld = DivBy(x, spDiv); if (!selIsOne) Ld = DivBy(Ld, selDiv); }
                                                else {
                                                    Ld = DivFast(x, spDiv.d, spDiv.r, trk);
                                                    if (!allSelOne) { const float Ld2 = DivFast(Ld, selDiv.d, selDiv.r, trk); Ld = selIsOne ? Ld : Ld2; }
                                                }
                                                o[k] = Get4(bt, k) * Ld;
                                            }
                                            if (!EXACT && __any(trk.Bad())) return false;
                                            tile.q[c][threadIdx.x] = make_float4(o[0], o[1], o[2], o[3]);
                                            accF |= aF;
                                            return true;
                                        };
#pragma unroll 1
                                        for (int c = 0; c < EXP_NQ; ++c)
                                            if (!quad(c, std::false_type{})) quad(c, std::true_type{});
                                    }
                                    fNonBlack |= accF != 0u;
                                } else {
                                    if (dark) {
#pragma unroll 1
                                        for (int c = 0; c < EXP_NQ; ++c) darkQuad(c, fQuad(c));
                                    }
                                    if (go && !dark) {
#pragma unroll 1
                                        for (int c = 0; c < EXP_NQ; ++c) misQuad(c, fQuad(c), isEnvLight ? zero4 : LoadSpec4(light.L, c));
                                    }
                                }
                                StoreSpectrumLines(tile, pool, Q_LMIS, slot, EXP_STORE(go && !dark));   // whole 128-B lines, as for the light sample
        STAMP(12);
                                if (fNonBlack && go) {
                                    if (dark) newFlags |= F_MIS_DARK;
                                    pool.R(R_MI0, slot) = make_float4(mr.o.x, mr.o.y, mr.o.z, mr.d.x);
                                    pool.R(R_MI1, slot) = make_float4(mr.d.y, mr.d.z, misTHi, misWord);
                                    if (s.nLights > 1) pool.I(I_MISLIGHT, slot) = lightNum;   // (one light: k_resolve_mis knows which)
                                    newFlags |= F_MIS;
                                }
                            }
                        }
                    }
                }
            }
            // ---- sample the BSDF for the next direction, path.cpp:131-150
            {
                V3 wo = -rd, wi;
                float pdf = 0;
                int sflags = 0;
                float u2[2];
                if constexpr (FUSED_HALTON) { ScrambledDimensionsFused<2>(s.primes, s.primeSums, s.perms, s.primeMagic, ps.index, ps.dim, u2); ps.dim += 2; }
                else Get2D<HALTON_ONLY>(s, ps, pixelPlane, samplePlane, slot, &u2[0], &u2[1]);
                const float u0 = u2[0], u1 = u2[1];
                BSDFEvalT<NL> ev;
                V3 wiLocal;
                const bool ok = BSDF_Sample_f<NL, TM, !ACCUM>(fr, wo, &wi, u0, u1, &pdf, MI_BSDF_ALL, &sflags, &ev, &wiLocal);
                if constexpr (ACCUM) {
                    if (ok && pdf != 0.f) {
                        if (ev.n > 0) AccumulateSpecular<NL, TM>(ev.lobes[0], mat->bxdf, ltp, rdTile, wrTile);
                        else AccumulateFLocal<NL, TM>(fr, fr.WorldToLocal(wo), wiLocal, Dot(wi, fr.ng) * Dot(wo, fr.ng) > 0, MI_BSDF_ALL, ltp, rdTile, wrTile);
                    }
                }
                auto fQuad = [&](int c) -> float4 { if constexpr (ACCUM) return rdTile(c); else return EvalQuad<NL, TM>(ev, mat->bxdf, c, ltp); };
        STAMP(13);
                bool fNonBlack = false;
                if (ok && pdf != 0.f) {
                    const float absdot = AbsDot(wi, isect.shN);
                    float etaScale = ray1.w;
                    if ((sflags & MI_BSDF_SPECULAR) && (sflags & MI_BSDF_TRANSMISSION)) {
                        float eta = mat->eta;
                        etaScale *= (Dot(wo, isect.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                    }
                    const Divisor pdfDiv = MakeDivisor(pdf);
                    float maxRR = 0;
                    auto contQuad = [&](int c, const float4 &fq) {  // beta *= f * |wi.ns| / pdf (only meaningful when f is not black)
                        float4 bt = loadBeta(c);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int b = 4 * c + k;
                            if (b < MI_NSPEC) {
                                const float f = Get4(fq, k);
                                fNonBlack |= (f != 0.f);
                                const float nb = Get4(bt, k) * DivBy(f * absdot, pdfDiv);
                                Set4(bt, k, nb);
                                const float rr = nb * etaScale;
                                maxRR = (b == 0) ? rr : maxf(maxRR, rr);
                            }
                        }
                        tile.q[c][threadIdx.x] = bt;
                    };
                    if constexpr (SIMPLE_SPECTRA) {
                        const SimpleLobes<NL> sl = MakeSimpleLobes<NL, TM>(ev);
                        const mi_bxdf *bx = mat->bxdf;
                        const bool divFast = sl.divisorsFast && pdfDiv.fast;
                        unsigned accF = 0u;
                        auto quad = [&](int c, auto exactTag) {
                            constexpr bool EXACT = decltype(exactTag)::value;
                            DivTrack trk;
                            trk.Reset(divFast);
                            const float4 bt = loadBeta(c);
                            float4 fq = SimpleEvalQuad<NL, TM, EXACT>(sl, bx, c, trk);
                            if (c == NQ - 1) fq.w = 0.f;
                            unsigned aF = 0u;
                            float o[4], mr = maxRR;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float f = Get4(fq, k);
                                OrNonZero(aF, f);
                                const float x = f * absdot;
                                o[k] = Get4(bt, k) * (EXACT ? DivBy(x, pdfDiv) : DivFast(x, pdfDiv.d, pdfDiv.r, trk));
                                const float rr = o[k] * etaScale;
                                if (k == 0) mr = (c == 0) ? rr : maxf(mr, rr);
                                else if (k < 3) mr = maxf(mr, rr);
                                else mr = (c == NQ - 1) ? mr : maxf(mr, rr);
                            }
                            if (c == NQ - 1) o[3] = bt.w;   // (the pad word keeps what it held)
                            if (!EXACT && __any(trk.Bad())) return false;
                            tile.q[c][threadIdx.x] = make_float4(o[0], o[1], o[2], o[3]);
                            accF |= aF;
                            maxRR = mr;
                            return true;
                        };
#pragma unroll 1
                        for (int c = 0; c < EXP_NQ; ++c)
                            if (!quad(c, std::false_type{})) quad(c, std::true_type{});
                        fNonBlack |= accF != 0u;
                    } else {
#pragma unroll 1
                        for (int c = 0; c < EXP_NQ; ++c) contQuad(c, fQuad(c));
                    }
                    // (the new throughput is stored below, once it is known that a later vertex will read it)
                    bool killed = false;
        STAMP(14);
                    if (fNonBlack) {
                        // Russian roulette, path.cpp:176-184
                        if (maxRR < s.rrThreshold && bounces > 3) {
                            float q = maxf(.05f, 1 - maxRR);
                            if (Get1D<HALTON_ONLY>(s, ps, pixelPlane, samplePlane, slot) < q) killed = true;
                            else {
                                const Divisor inv = MakeDivisor(1 - q);
#pragma unroll 1
                                for (int c = 0; c < NQ; ++c) {
                                    float4 bt = tile.q[c][threadIdx.x];   // (the new beta, parked in the tile)
#pragma unroll
                                    for (int k = 0; k < 4; ++k)
                                        if (4 * c + k < MI_NSPEC) Set4(bt, k, DivBy(Get4(bt, k), inv));
                                    tile.q[c][threadIdx.x] = bt;
                                }
                            }
                        }
                    }
                    // the next vertex reads the throughput if it is shaded (bounces + 1 < maxDepth) or may show emitted
                    // light (after a specular bounce, path.cpp:91-101); a path that ends here, or whose last ray only
                    // has to be traced, leaves none behind
                    const bool needBeta = fNonBlack && !killed && (bounces + 1 < s.maxDepth || (sflags & MI_BSDF_SPECULAR));
                    StoreSpectrumLines(tile, pool, Q_BETA, slot, EXP_STORE(needBeta));
        STAMP(15);
                    if (needBeta) betaWritten = true;
                    if (fNonBlack) {
                        if (killed) finished = true;
                        else {
                            const Ray nr = SpawnRay(isect, wi);
                            pool.R(R_RAY0, slot) = make_float4(nr.o.x, nr.o.y, nr.o.z, nr.tMax);
                            pool.R(R_RAY1, slot) = make_float4(nr.d.x, nr.d.y, nr.d.z, etaScale);
                            if (sflags & MI_BSDF_SPECULAR) newFlags |= F_SPECULAR;
                        }
                    }
                }
                if (!(ok && pdf != 0.f && fNonBlack)) finished = true;
            }
        STAMP(16);
            dimNow = ps.dim;
            if constexpr (!HALTON_ONLY) { if (IsPixelSampler(s)) { pool.I(I_DIM, slot) = ps.dim; dimNow = 0; } }
            if (!HALTON_ONLY && s.samplerType >= MI_SAMPLER_RANDOM) {   // the stream moves on with the path
                pool.I(I_IDXLO, slot) = (int)(uint32_t)ps.index;
                pool.I(I_IDXHI, slot) = (int)(uint32_t)(ps.index >> 32);
            }
        }
        if (!passThrough) {
            if (finished) {
                // ReportValue(pathLength, bounces): `bounces` at the break of path.cpp's loop
                pathLen = (unsigned)bounces;
                newFlags = (newFlags & (F_NEE | F_SHADOW | F_MIS | F_CAND | F_NEE_NZ | F_MIS_DARK)) | F_FINISHED;
            } else {
                newFlags |= F_ALIVE;
            }
            if (lZero) newFlags |= F_L_ZERO;
            newFlags |= flags & F_L_IN_B;
            if (betaOne && !betaWritten) newFlags |= F_BETA_ONE;
            wantShadow = (newFlags & F_SHADOW) != 0;
            wantMis = (newFlags & F_MIS) != 0;
            // a direct-lighting estimate with neither ray pending is already known to be black
            if ((newFlags & F_NEE) && !wantShadow && !wantMis) { ++zeroNow; newFlags &= ~F_NEE; }
            pool.I(I_FLAGS, slot) = StateWord(newFlags, finished ? bounces : bounces + 1, dimNow);
        }
    }
    STAMP(17);
    __shared__ unsigned sAppend[10];
    unsigned posS, posM;
    BlockReserve2(&ctr->shadowCount.v, wantShadow, &ctr->misCount.v, wantMis, sAppend, &posS, &posM);
    if (wantShadow) pool.shadowQ[posS] = slot;
    if (wantMis) pool.misQ[posM] = slot;
    {   // the three statistics of a wave in one reduction: per lane at most one path, one black estimate and 255 bounces
        unsigned packed = totalPaths | (zeroNow << 8) | (pathLen << 16);
        for (int off = 32; off > 0; off >>= 1) packed += __shfl_down(packed, off, 64);
        if ((threadIdx.x & 63) == 0 && packed) {
            DevStats &st8 = Stats(ctr);
            if (packed & 0xffu) atomicAdd(&st8.totalPaths, (unsigned long long)(packed & 0xffu));
            if ((packed >> 8) & 0xffu) atomicAdd(&st8.zeroRadiancePaths, (unsigned long long)((packed >> 8) & 0xffu));
            if (packed >> 16) atomicAdd(&st8.pathLengthSum, (unsigned long long)(packed >> 16));
        }
    }
    STAMP(18);
#ifdef MIPT_EXP_STAMPS
    if (stampOn && (threadIdx.x & 63) == 0) atomicAdd(&ctr->phaseWaves, 1ull);
#endif
}

#if MIPT_HAS_MAIN
// ------------------------------------------------------------------ spatial light distribution (create time)
__global__ void k_build_spatial(DScene s, float *func, float *cdf, float *funcInt, uint32_t nVox) {
    const uint32_t vox = blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= nVox) return;
    const int nL = (int)s.nLights;
    int pi[3];
    pi[0] = vox % s.nVoxels[0];
    pi[1] = (vox / s.nVoxels[0]) % s.nVoxels[1];
    pi[2] = vox / (s.nVoxels[0] * s.nVoxels[1]);
    float lo[3], hi[3];
    for (int i = 0; i < 3; ++i) {
        float t0 = (float)pi[i] / (float)s.nVoxels[i], t1 = (float)(pi[i] + 1) / (float)s.nVoxels[i];
        float a = lerpf(t0, s.wbMin[i], s.wbMax[i]), b = lerpf(t1, s.wbMin[i], s.wbMax[i]);
        lo[i] = minf(a, b); hi[i] = maxf(a, b);
    }
    float *f = func + (size_t)vox * nL;
    for (int j = 0; j < nL; ++j) f[j] = 0.f;
    const int nSamples = 128;
    for (int i = 0; i < nSamples; ++i) {
        float t[3] = {RadicalInverse(s, 0, i), RadicalInverse(s, 1, i), RadicalInverse(s, 2, i)};
        Interaction intr;
        intr.p = V3(lerpf(t[0], lo[0], hi[0]), lerpf(t[1], lo[1], hi[1]), lerpf(t[2], lo[2], hi[2]));
        intr.wo = V3(1, 0, 0);
        float u0 = RadicalInverse(s, 3, i), u1 = RadicalInverse(s, 4, i);
        for (int j = 0; j < nL; ++j) {
            const mi_light &l = s.lights[j];
            LightSample ls = SampleLi<TM_ALL>(s, l, intr, u0, u1);
            if (ls.pdf > 0) {
                float yy = 0.f;
                if (!ls.black) for (int b = 0; b < MI_NSPEC; ++b) yy += s.cieY[b] * LiBin<TM_ALL>(s, l, ls, b);
                f[j] += YScale(yy) / ls.pdf;
            }
        }
    }
    float sumContrib = 0;
    for (int j = 0; j < nL; ++j) sumContrib += f[j];
    float avgContrib = sumContrib / (float)((size_t)nSamples * (size_t)nL);
    float minContrib = (avgContrib > 0) ? (float)(.001 * (double)avgContrib) : 1.f;
    for (int j = 0; j < nL; ++j) f[j] = maxf(f[j], minContrib);
    // Distribution1D ctor, sampling.h:57-70
    float *c = cdf + (size_t)vox * (nL + 1);
    c[0] = 0;
    for (int j = 1; j < nL + 1; ++j) c[j] = c[j - 1] + f[j - 1] / nL;
    float fi = c[nL];
    if (fi == 0) { for (int j = 1; j < nL + 1; ++j) c[j] = (float)j / (float)nL; }
    else { for (int j = 1; j < nL + 1; ++j) c[j] /= fi; }
    funcInt[vox] = fi;
}

// ------------------------------------------------------------------ pixel samplers' tables (create time)
// ZeroTwoSequenceSampler::StartPixel (zerotwosequence.cpp:53-69) / StratifiedSampler::StartPixel (stratified.cpp:43-58) for every
// pixel at once, one lane per pixel, each with the pixel's own PCG32 stream (mi_sampler_type): all 1D tables, then all 2D
// tables, each scrambled / jittered and shuffled in the reference's order of RNG draws (VanDerCorput / Sobol2D,
// lowdiscrepancy.h:149-227; StratifiedSample1D / 2D, sampling.cpp:42-60; Shuffle, sampling.h:150-157).
DEV uint32_t PcgBounded(uint64_t &state, uint64_t inc, uint32_t b) {   // RNG::UniformUInt32(b), rng.h:112-121
    const uint32_t threshold = (~b + 1u) % b;
    while (true) {
        const uint32_t r = PcgNext(state, inc);
        if (r >= threshold) return r % b;
    }
}
__global__ void k_pixel_tables(DScene s, float *tab1, float *tab2, unsigned long long nPix) {
    const unsigned long long pix = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= nPix) return;
    const int spp = (int)s.samplesPerPixel, dims = s.pixelDims;
    const uint64_t inc = (((uint64_t)pix) << 1u) | 1u;
    uint64_t state = RandomStreamStart(inc);
    const bool zeroTwo = s.samplerType == MI_SAMPLER_ZEROTWO;
    for (int d = 0; d < dims; ++d) {   // the 1D tables
        float *v = tab1 + ((size_t)pix * dims + d) * (size_t)spp;
        if (zeroTwo) {   // VanDerCorput(1, spp, ...): the generator matrix is the bit-reversed identity, C[k] = 2^(31 - k)
            uint32_t x = PcgNext(state, inc);
            for (int i = 0; i < spp; ++i) {
                v[i] = minf((float)x * 0x1p-32f, kOneMinusEpsilon);
                x ^= 0x80000000u >> __builtin_ctz((unsigned)(i + 1));
            }
            for (int i = 0; i < spp; ++i) (void)PcgBounded(state, inc, 1u);   // Shuffle of each sample's single value: one draw apiece
        } else {
            const float invN = 1.f / (float)spp;
            for (int i = 0; i < spp; ++i) {
                const float delta = s.jitter ? PcgFloat(state, inc) : 0.5f;
                v[i] = minf(((float)i + delta) * invN, kOneMinusEpsilon);
            }
        }
        for (int i = 0; i < spp; ++i) {   // Shuffle(v, spp, 1, rng)
            const int other = i + (int)PcgBounded(state, inc, (uint32_t)(spp - i));
            const float t = v[i]; v[i] = v[other]; v[other] = t;
        }
    }
    for (int d = 0; d < dims; ++d) {   // the 2D tables
        float2 *v = reinterpret_cast<float2 *>(tab2) + ((size_t)pix * dims + d) * (size_t)spp;
        if (zeroTwo) {   // Sobol2D(1, spp, ...): second matrix = Pascal's triangle mod 2, column k = column k-1 ^ (column k-1 >> 1)
            uint32_t x0 = PcgNext(state, inc), x1 = PcgNext(state, inc);
            uint32_t col[32];
            col[0] = 0x80000000u;
            for (int k = 1; k < 32; ++k) col[k] = col[k - 1] ^ (col[k - 1] >> 1);
            for (int i = 0; i < spp; ++i) {
                v[i] = make_float2(minf((float)x0 * 0x1p-32f, kOneMinusEpsilon), minf((float)x1 * 0x1p-32f, kOneMinusEpsilon));
                const int k = __builtin_ctz((unsigned)(i + 1));
                x0 ^= 0x80000000u >> k;
                x1 ^= col[k];
            }
            for (int i = 0; i < spp; ++i) (void)PcgBounded(state, inc, 1u);
        } else {
            const float dx = 1.f / (float)s.xSamples, dy = 1.f / (float)s.ySamples;
            int i = 0;
            for (int y = 0; y < s.ySamples; ++y)
                for (int x = 0; x < s.xSamples; ++x, ++i) {
                    const float jx = s.jitter ? PcgFloat(state, inc) : 0.5f;
                    const float jy = s.jitter ? PcgFloat(state, inc) : 0.5f;
                    v[i] = make_float2(minf(((float)x + jx) * dx, kOneMinusEpsilon), minf(((float)y + jy) * dy, kOneMinusEpsilon));
                }
        }
        for (int i = 0; i < spp; ++i) {
            const int other = i + (int)PcgBounded(state, inc, (uint32_t)(spp - i));
            const float2 t = v[i]; v[i] = v[other]; v[other] = t;
        }
    }
}

// ------------------------------------------------------------------ texture lookups on their own (mi_pt_texture_lookup)
__global__ void k_texture_lookup(DScene s, int tex, const float *q, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mi_texture &t = s.textures[tex];
    const RGB3 v = MipLookup(s, s.mipmaps[t.mipmap], q[6 * i], q[6 * i + 1], q[6 * i + 2], q[6 * i + 3], q[6 * i + 4], q[6 * i + 5], t.filter, t.max_aniso);
    out[3 * i] = v.r; out[3 * i + 1] = v.g; out[3 * i + 2] = v.b;
}

// ------------------------------------------------------------------ scalar helpers on their own (mi_pt_math_probe)
__global__ void k_math_probe(int op, uint32_t n, const float *x, const float *y, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1], y0 = y[2 * i], y1 = y[2 * i + 1];
    float *o = out + 3 * (size_t)i;
    o[0] = o[1] = o[2] = 0.f;
    if (op == 0) { o[0] = NextFloatUp(x0); o[1] = NextFloatDown(x0); }
    else if (op >= 1 && op <= 4) {
        const EFloat a(x0, x1), b(y0, y1);
        const EFloat r = op == 1 ? a + b : (op == 2 ? a - b : (op == 3 ? a * b : a / b));
        o[0] = r.v; o[1] = r.low; o[2] = r.high;
    } else if (op == 5) {
        const float cdf[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9}, func[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
        float pdf;
        o[0] = (float)SampleDiscrete(func, cdf, 1.f, 9, x0, &pdf);
    }
}

// ------------------------------------------------------------------ standalone traversal (mi_pt_trace)
__global__ void __launch_bounds__(BLOCK) k_trace(DScene s, const float *rays, uint32_t n, int anyHit, float *hits) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + (size_t)i * 7;
    V3 ro(r[0], r[1], r[2]), rd(r[3], r[4], r[5]);
    unsigned nodes = 0, tris = 0;
    Hit h;
    h.prim = -1; h.t = 0; h.b0 = h.b1 = h.b2 = 0;
    int prim;
    if (anyHit) prim = Traverse<true, true>(s, ro, rd, r[6], &h, nodes, tris) ? 0 : -1;
    else prim = Traverse<false, true>(s, ro, rd, r[6], &h, nodes, tris) ? h.prim : -1;
    float *o = hits + (size_t)i * 4;
    o[0] = __int_as_float(prim);
    o[1] = (prim >= 0 && !anyHit) ? h.t : 0.f;
    o[2] = (prim >= 0 && !anyHit) ? h.b0 : 0.f;
    o[3] = (prim >= 0 && !anyHit) ? h.b1 : 0.f;
}

// ------------------------------------------------------------------ recorded rays through the render's own kernels (mi_pt_trace_wavefront)
// Loads ray i into slot i of the pool exactly as k_generate / k_shade leave a path ray (mode 0), an NEE shadow ray (1) or a
// BSDF-sampled MIS ray (2), and lists it in that mode's work list. The shadow mode's flags make k_resolve_shadow's commit
// readable without a spectrum: the candidate line "holds L + the light sample" (F_CAND), so an unoccluded ray flips F_L_IN_B and clears F_L_ZERO.
__global__ void __launch_bounds__(BLOCK) k_trace_load(Pool pool, DevCounters *ctr, const float *rays, uint32_t n, int mode) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) {
        if (mode == 0) { ctr->primCount.v = n; ctr->contCount.v = 0; }
        else if (mode == 1) ctr->shadowCount.v = n;
        else ctr->misCount.v = n;
    }
    if (i >= pool.n) return;
    int flags = 0;
    if (i < n) {
        const float *r = rays + (size_t)i * 7;
        if (mode == 0) {
            pool.R(R_RAY0, i) = make_float4(r[0], r[1], r[2], r[6]);
            pool.R(R_RAY1, i) = make_float4(r[3], r[4], r[5], 1.f);
            pool.extQ[i] = i;
            flags = F_ALIVE;
        } else {
            pool.R(mode == 1 ? R_SH0 : R_MI0, i) = make_float4(r[0], r[1], r[2], r[3]);
            // (mode 3, the visibility form of a MIS ray: the span ends at the given tMax and starts at tMax (1 - 2^-8); no primitive left out)
            pool.R(mode == 1 ? R_SH1 : R_MI1, i) = make_float4(r[4], r[5], mode == 3 ? r[6] : 0.f, mode == 3 ? __uint_as_float(MIS_EXCL_NONE | (8u << MIS_EXCL_BITS)) : 0.f);
            (mode == 1 ? pool.shadowQ : pool.misQ)[i] = i;
            flags = mode == 1 ? (F_ALIVE | F_NEE | F_SHADOW | F_L_ZERO | F_CAND | F_NEE_NZ) : (F_ALIVE | F_NEE | F_MIS);
        }
        pool.I(I_HITPRIM, i) = -2;   // (every ray must be answered: k_trav overwrites this)
        if (mode == 1) pool.shadowQ[pool.n + i] = 0xfffffffeu;   // (... a shadow or MIS ray's answer lies beside its queue entry)
        if (mode >= 2) { pool.misQ[pool.n + 2 * (size_t)i] = mode == 3 ? 0xfffffffdu : 0xfffffffeu; pool.misQ[pool.n + 2 * (size_t)i + 1] = 0u; }
        pool.I(I_NPEND, i) = 0;
        pool.I(I_HITINST, i) = -1;
        pool.I(I_MISLIGHT, i) = 0;
    }
    pool.I(I_FLAGS, i) = flags;
}
// what k_trav left in the planes, before the resolve step
__global__ void __launch_bounds__(BLOCK) k_trace_raw(Pool pool, uint32_t n, int mode, float *extra) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (mode == 1) {   // the answer word: -2 = never answered; else bit 31 occluded | postponed quadrics
        const unsigned v = pool.shadowQ[pool.n + i];
        extra[4 * (size_t)i + 2] = __int_as_float(v == 0xfffffffeu ? 0 : (int)(v & 0x7fffffffu));
        extra[4 * (size_t)i + 3] = __int_as_float(v == 0xfffffffeu ? -2 : ((v >> 31) ? 0 : -1));
        return;
    }
    if (mode >= 2) {
        extra[4 * (size_t)i + 2] = __int_as_float((int)pool.misQ[pool.n + 2 * (size_t)i + 1]);
        extra[4 * (size_t)i + 3] = __int_as_float((int)pool.misQ[pool.n + 2 * (size_t)i]);
        return;
    }
    extra[4 * (size_t)i + 2] = __int_as_float(pool.I(I_NPEND, i));
    extra[4 * (size_t)i + 3] = __int_as_float(pool.I(I_HITPRIM, i));
}
// The committed answer. Mode 0: the planes as k_resolve_extend / k_resolve_overflow left them. Mode 1: k_resolve_shadow's
// verdict (see k_trace_load). Mode 2: k_resolve_mis consumes its hit in place (ResolveMisSlot), so the quadric step is run
// here with the same device function and arguments it uses.
template <bool INST>
__global__ void __launch_bounds__(BLOCK) k_trace_read(DScene s, Pool pool, uint32_t n, int mode, float *hits, float *extra) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    int prim = mode >= 2 ? (int)pool.misQ[pool.n + 2 * (size_t)i] : pool.I(I_HITPRIM, i), inst = -1;
    float4 hr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 0) { hr = pool.R(R_HIT, i); if (INST) inst = pool.I(I_HITINST, i); }
    else if (mode == 1) prim = (pool.I(I_FLAGS, i) & F_L_ZERO) ? 0 : -1;
    else if (mode == 3) {}   // (k_trav<3>'s verdict as it stands: an occluder, -1, -2 = ambiguous; -3 = never answered)
    else {
        const float4 r0 = pool.R(R_MI0, i), r1 = pool.R(R_MI1, i);
        const V3 ro(r0.x, r0.y, r0.z), rd(r0.w, r1.x, r1.y);
        hr = pool.R(R_HIT, i);
        Hit h;
        h.prim = prim; h.t = hr.x; h.b0 = hr.y; h.b1 = hr.z; h.b2 = hr.w;
        bool found = prim >= 0;
        unsigned nodes = 0, tris = 0;
        const int npend = (int)pool.misQ[pool.n + 2 * (size_t)i + 1];
        if (npend & PEND_OVERFLOW) found = ResolveQuadrics<false, INST, true>(s, pool, i, ro, rd, kInfinity, &h, found, nodes, tris);
        else if (npend != 0) found = ResolveQuadrics<false, INST, false>(s, pool, i, ro, rd, kInfinity, &h, found, nodes, tris, npend);
        prim = found ? h.prim : -1;
        hr = make_float4(h.t, h.b0, h.b1, h.b2);
        inst = h.inst;
    }
    const bool rec = prim >= 0 && mode != 1 && mode != 3;
    float *o = hits + (size_t)i * 4;
    o[0] = __int_as_float(prim);
    o[1] = rec ? hr.x : 0.f; o[2] = rec ? hr.y : 0.f; o[3] = rec ? hr.z : 0.f;
    if (extra) { extra[4 * (size_t)i] = rec ? hr.w : 0.f; extra[4 * (size_t)i + 1] = __int_as_float(rec ? inst : -1); }
}

__global__ void k_film_split(const float *film32, float *filmSum, float *weightSum, size_t nPix) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPix * 32) return;
    size_t pix = i >> 5;
    int b = (int)(i & 31);
    float v = film32[i];
    if (b < 31) { if (filmSum) filmSum[pix * 31 + b] = v; }
    else if (weightSum) weightSum[pix] = v;
}
#endif   // MIPT_HAS_MAIN

// The k_shade instances LaunchShade uses, in three groups of about equal compile time.
constexpr unsigned TM_FULL = TM_ALL & ~TM_INSTANCES, TM_GENERIC = TM_FULL & ~TM_TEXTURED;
#define MIPT_SHADE_GROUP_1(X) X(MI_MAX_BXDFS, TM_ALL) X(4, TM_GENERIC) X(2, TM_GENERIC) \
    X(2, TM_DIFFUSE | TM_LIGHTS_ALL) X(2, TM_DIFFUSE | TM_LIGHTS_NO_ENV) X(2, TM_DIFFUSE | TM_LIGHTS_ALL | TM_SAMPLERS) X(2, TM_DIFFUSE | TM_LIGHTS_NO_ENV | TM_SAMPLERS)
#define MIPT_SHADE_GROUP_2(X) X(MI_MAX_BXDFS, TM_FULL) X(2, TM_ALL) X(2, TM_FULL) \
    X(2, TM_PLASTIC | TM_LIGHTS_ALL) X(2, TM_PLASTIC | TM_LIGHTS_NO_ENV) X(2, TM_PLASTIC | TM_LIGHTS_ALL | TM_SAMPLERS) X(2, TM_PLASTIC | TM_LIGHTS_NO_ENV | TM_SAMPLERS)
#define MIPT_SHADE_GROUP_3(X) X(MI_MAX_BXDFS, TM_GENERIC) X(4, TM_FULL) X(2, TM_GLASS | TM_LIGHTS_ALL | TM_SAMPLERS) X(4, TM_UBER | TM_LIGHTS_ALL | TM_SAMPLERS) X(MI_MAX_BXDFS, TM_DISNEY | TM_LIGHTS_ALL | TM_SAMPLERS) \
    X(2, TM_DIFFUSE | TM_TEXTURED | TM_LIGHTS_ALL | TM_SAMPLERS) X(2, TM_PLASTIC | TM_TEXTURED | TM_LIGHTS_ALL | TM_SAMPLERS)
#define MIPT_SHADE_DEFINE(NL_, TM_) template __global__ void k_shade<NL_, (TM_)>(DScene, Pool, DevCounters *, unsigned);
#define MIPT_SHADE_EXTERN(NL_, TM_) extern template __global__ void k_shade<NL_, (TM_)>(DScene, Pool, DevCounters *, unsigned);
// MIPT_HOT_ONLY (tools/shade_experiments.sh): a one-translation-unit build with the matte and plastic Halton instances
// alone -- what the killeroo / Cornell-without-glass frames launch -- for quick same-box A/B runs of k_shade experiments.
#ifdef MIPT_PART
#if MIPT_PART == 0
MIPT_SHADE_GROUP_1(MIPT_SHADE_EXTERN) MIPT_SHADE_GROUP_2(MIPT_SHADE_EXTERN) MIPT_SHADE_GROUP_3(MIPT_SHADE_EXTERN)
#elif MIPT_PART == 1
MIPT_SHADE_GROUP_1(MIPT_SHADE_DEFINE)
#elif MIPT_PART == 2
MIPT_SHADE_GROUP_2(MIPT_SHADE_DEFINE)
#else
MIPT_SHADE_GROUP_3(MIPT_SHADE_DEFINE)
#endif
#endif

}  // namespace dptk
using namespace dptk;

#if MIPT_HAS_MAIN
// =============================================================================
// C ABI
// =============================================================================
constexpr int N_EV = 8;   // events per iteration: kernel-class boundaries + the start of the closest-hit traversal launch
struct SubRenderer {
    Pool pool{};
    DevCounters *ctr = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t evIter[2][N_EV] = {{nullptr}};  // per-iteration kernel boundaries, two alternating sets
    double t[7] = {0};
    int poolQuadPlanes = 0;   // spectral planes the pool was allocated with (spectralpath needs one set more)
    size_t poolBytes = 0;     // bytes of device memory behind the pool
    unsigned long long iterations = 0;
    DevCounters result{};
};

struct mi_pt {
    int device = 0;
    DScene scene{};
    std::vector<void *> allocs;
    float *film = nullptr;  // [nPix][32]
    float *stageSum = nullptr, *stageW = nullptr;   // [nPix][31] / [nPix] staging of the host hand-over
    size_t nPix = 0;
    int filmW = 0, filmH = 0;
    long long spp = 0;
    std::vector<SubRenderer> subs;
    double lastSeconds[8] = {0};
    unsigned smallClasses = 1u << MISS_CLASS, largeClasses = 0;  // shading classes with <= 2 lobes / with more
    uint32_t nTextures = 0;
    std::vector<int> textureTypes;
    bool hasAlphaMasks = false;      // picks the traversal kernels compiled with the alpha-mask test
    bool hasInstances = false;       // ... and with the TransformedPrimitive code (those carry the alpha-mask test too)
    bool hasQuadrics = false;        // the scene has spheres: the quadric lists of rays can overflow (k_resolve_overflow is launched)
    bool hasInfiniteLight = false;   // picks the kernels compiled with the environment-light code
    unsigned diffuseClasses = 0, plasticClasses = 0;
    unsigned texturedDiffuse = 0, texturedPlastic = 0;           // textured classes that fit the diffuse / plastic lobe masks
    unsigned glassClasses = 0;                                   // untextured glass / mirror lobe sets (<= 2 lobes)
    unsigned uberClasses = 0, disneyClasses = 0;                 // untextured uber-like (<= 4 lobes) and Disney lobe sets: instances of their own
    unsigned mediumClasses = 0, texturedMedium = 0;              // 3- and 4-lobe classes (uber with Kr / Kt, translucent): 4-lobe instances
    unsigned texturedSmall = 0, texturedLarge = 0;               // classes of image-textured materials (taken out of small / largeClasses)              // subsets of smallClasses run by the lobe-specialised kernels
    int numCUs = 256;
};

namespace {

template <typename T>
int Upload(mi_pt *pt, const T *src, size_t count, const T **dst) {
    *dst = nullptr;
    if (count == 0 || !src) {
        void *p = nullptr;
        HIPCHK(hipMalloc(&p, 16));
        pt->allocs.push_back(p);
        *dst = (const T *)p;
        return MI_OK;
    }
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, count * sizeof(T) + 16));   // (16 B of slack: the kernels read spectra as 16-B quads)
    pt->allocs.push_back(p);
    HIPCHK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T *)p;
    return MI_OK;
}

// Device bytes of one path slot: planes, records, spectra and its entries in the queues.
size_t PoolSlotBytes(int nQuadPlanes) {
    return (size_t)P_COUNT * sizeof(float) + (size_t)nQuadPlanes * sizeof(float4) + (size_t)R_COUNT * sizeof(float4) + (size_t)I_COUNT * sizeof(int) +
           (size_t)(6 + MAX_CLASSES + 3) * sizeof(uint32_t);
}

void FreePool(Pool &p) {
    hipFree(p.f); hipFree(p.q); hipFree(p.r); hipFree(p.i); hipFree(p.shadowQ); hipFree(p.extQ); hipFree(p.misQ); hipFree(p.shadeQ); hipFree(p.ovfQ);
    p = Pool{};   // n = 0, every pointer null: a later render cannot mistake a half-built pool for a usable one
}

// Failure-safe: the new pool is built aside and swapped in only when every allocation succeeded; after a failure the
// sub-renderer holds no pool at all (n == 0), so the next render allocates afresh instead of launching on stale sizes.
int EnsurePool(SubRenderer &sub, uint32_t n, int nQuadPlanes) {
    Pool &p = sub.pool;
    if (p.n == n && p.f && sub.poolQuadPlanes == nQuadPlanes) return MI_OK;
    FreePool(p);
    sub.poolQuadPlanes = 0;
    sub.poolBytes = 0;
    Pool t{};
    const bool ok = hipMalloc((void **)&t.f, (size_t)P_COUNT * n * sizeof(float)) == hipSuccess &&
                    hipMalloc((void **)&t.q, (size_t)nQuadPlanes * n * sizeof(float4)) == hipSuccess &&
                    hipMalloc((void **)&t.r, (size_t)R_COUNT * n * sizeof(float4)) == hipSuccess &&
                    hipMalloc((void **)&t.i, (size_t)I_COUNT * n * sizeof(int)) == hipSuccess &&
                    hipMalloc((void **)&t.shadowQ, (size_t)2 * n * sizeof(uint32_t)) == hipSuccess &&
                    hipMalloc((void **)&t.extQ, (size_t)n * sizeof(uint32_t)) == hipSuccess &&
                    hipMalloc((void **)&t.misQ, (size_t)3 * n * sizeof(uint32_t)) == hipSuccess &&
                    hipMalloc((void **)&t.shadeQ, (size_t)MAX_CLASSES * n * sizeof(uint32_t)) == hipSuccess &&
                    hipMalloc((void **)&t.ovfQ, (size_t)3 * n * sizeof(uint32_t)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();   // the failed hipMalloc must not poison the next call's hipGetLastError
        FreePool(t);
        g_err = "hipMalloc(path pool of " + std::to_string(n) + " slots) failed";
        return MI_ERR_NOMEM;
    }
    t.n = n;
    p = t;
    sub.poolQuadPlanes = nQuadPlanes;
    sub.poolBytes = (size_t)n * PoolSlotBytes(nQuadPlanes);
    return MI_OK;
}

// Device temporaries of the entry points: freed on every return path.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <typename T> T *as() const { return (T *)p; }
};

}  // namespace

extern "C" {

const char *mi_pt_last_error(void) { return g_err.c_str(); }

int mi_pt_create(const mi_scene_desc *d, int device_ordinal, mi_pt **out) {
    if (!d || !out) { g_err = "null argument"; return MI_ERR_INVALID; }
    if (d->abi_version != MI_ABI_VERSION) { g_err = "mi_scene_desc ABI version mismatch"; return MI_ERR_INVALID; }
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev == 0) { g_err = "no HIP device available (this path has no CPU fallback)"; return MI_ERR_NO_DEVICE; }
    if (device_ordinal < 0 || device_ordinal >= nDev) { g_err = "device ordinal out of range"; return MI_ERR_NO_DEVICE; }
    // host-side shape checks before anything is launched
    if (d->n_prims && !d->n_nodes) { g_err = "primitives without BVH nodes"; return MI_ERR_INVALID; }
    for (uint32_t i = 0; i < d->n_nodes; ++i) {
        const mi_bvh_node &n = d->nodes[i];
        if (n.n_prims > 0) { if (n.offset < 0 || (uint32_t)n.offset + n.n_prims > d->n_prims) { g_err = "BVH leaf out of range"; return MI_ERR_INVALID; } }
        else if (n.offset <= (int)i || (uint32_t)n.offset >= d->n_nodes || i + 1 >= d->n_nodes) { g_err = "BVH child out of range"; return MI_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const mi_prim &p = d->prims[i];
        if (p.instance != 0) {
            if (p.instance < 0 || (uint32_t)p.instance > d->n_instances || !d->instances) { g_err = "primitive instance index out of range"; return MI_ERR_INVALID; }
            if (d->instances[p.instance - 1].root >= d->n_nodes) { g_err = "instance BVH root out of range"; return MI_ERR_INVALID; }
            continue;
        }
        if (p.shape >= 0 ? (uint32_t)p.shape >= d->n_tris : (uint32_t)(~p.shape) >= d->n_spheres) { g_err = "primitive shape index out of range"; return MI_ERR_INVALID; }
        if (p.material >= (int)d->n_materials || p.area_light >= (int)d->n_lights) { g_err = "primitive material/light index out of range"; return MI_ERR_INVALID; }
    }
    for (uint32_t i = 0; i < d->n_tris * 3; ++i)
        if (d->tri_indices[i] < 0 || (uint32_t)d->tri_indices[i] >= d->n_verts) { g_err = "triangle vertex index out of range"; return MI_ERR_INVALID; }
    for (uint32_t i = 0; i < d->n_tris; ++i)
        if (d->tri_mesh[i] >= d->n_meshes) { g_err = "triangle mesh index out of range"; return MI_ERR_INVALID; }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->materials[i].n_bxdfs < 0 || d->materials[i].n_bxdfs > MI_MAX_BXDFS) { g_err = "material lobe count out of range"; return MI_ERR_INVALID; }
        if (d->materials[i].bump_tex >= (int)d->n_textures || d->materials[i].bump_tex < -1) { g_err = "mi_material.bump_tex out of range"; return MI_ERR_INVALID; }
        for (int k = 0; k < 2; ++k) {   // roughness maps: float image textures with a pyramid behind them (EvalFloatImageTexture)
            const int rt = d->materials[i].rough_tex[k];
            if (rt < -1 || rt >= (int)d->n_textures) { g_err = "mi_material.rough_tex out of range"; return MI_ERR_INVALID; }
            if (rt >= 0 && (d->textures[rt].type != MI_TEX_IMAGEMAP || (uint32_t)d->textures[rt].mipmap >= d->n_mipmaps)) {
                g_err = "mi_material.rough_tex must name an image texture with a mipmap"; return MI_ERR_INVALID;
            }
        }
        {   // (ABI v10) the sigma map, the lobe rules and the glass switch: what the shading kernels index with them
            const mi_material &m = d->materials[i];
            if (m.sigma_tex < -1 || m.sigma_tex >= (int)d->n_textures ||
                (m.sigma_tex >= 0 && (d->textures[m.sigma_tex].type != MI_TEX_IMAGEMAP || (uint32_t)d->textures[m.sigma_tex].mipmap >= d->n_mipmaps))) {
                g_err = "mi_material.sigma_tex must be -1 or name an image texture with a mipmap"; return MI_ERR_INVALID;
            }
            for (int k = 0; k < m.n_bxdfs; ++k) {
                const mi_lobe_tex &lt = m.tex[k];
                if (lt.rule < MI_LOBE_IF_R || lt.rule > MI_LOBE_METAL) { g_err = "mi_lobe_tex.rule is not a mi_lobe_rule"; return MI_ERR_INVALID; }
                if (lt.tex_R < -1 || lt.tex_R >= (int)d->n_textures || lt.tex_S < -1 || lt.tex_S >= (int)d->n_textures) { g_err = "mi_lobe_tex texture index out of range"; return MI_ERR_INVALID; }
                if (lt.rule >= MI_LOBE_DISNEY_SHEEN && lt.rule <= MI_LOBE_DISNEY_STRANS && lt.tex_R < 0) { g_err = "a \"disney\" lobe rule needs the colour's texture in tex_R"; return MI_ERR_INVALID; }
            }
            if ((m.rough_flags & MI_ROUGH_GLASS) && (m.n_bxdfs < 1 || m.bxdf[0].type != MI_BXDF_FRESNEL_SPECULAR)) {
                g_err = "MI_ROUGH_GLASS: lobe 0 must be the FresnelSpecular one"; return MI_ERR_INVALID;
            }
        }
    }
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        const mi_light &l = d->lights[i];
        if (l.type == MI_LIGHT_DIFFUSE_AREA && (l.shape >= 0 ? (uint32_t)l.shape >= d->n_tris : (uint32_t)(~l.shape) >= d->n_spheres)) { g_err = "area light shape index out of range"; return MI_ERR_INVALID; }
    }
    if (d->integrator.n_ca_bands < 1 || d->integrator.n_ca_bands > MI_NSPEC) { g_err = "n_ca_bands must be in [1, 31]"; return MI_ERR_INVALID; }
    if (d->integrator.max_depth < 0 || d->integrator.max_depth > 255) { g_err = "max_depth must be in [0, 255] (a path's bounce count travels in 8 bits of its state word)"; return MI_ERR_UNSUPPORTED; }
    if (d->sampler.type < MI_SAMPLER_HALTON || d->sampler.type > MI_SAMPLER_STRATIFIED) { g_err = "unknown mi_sampler.type"; return MI_ERR_INVALID; }
    if (d->sampler.samples_per_pixel < 1) { g_err = "mi_sampler.samples_per_pixel must be positive"; return MI_ERR_INVALID; }
    if (d->sampler.type >= MI_SAMPLER_ZEROTWO) {
        const mi_sampler &sm = d->sampler;
        if (sm.pixel_dims < 0 || sm.pixel_dims > 64) { g_err = "mi_sampler.pixel_dims must be in [0, 64]"; return MI_ERR_INVALID; }
        if (sm.samples_per_pixel > (1 << 20)) { g_err = "pixel samplers tabulate every sample of a pixel: at most 2^20 samples per pixel"; return MI_ERR_UNSUPPORTED; }
        if (sm.type == MI_SAMPLER_STRATIFIED && (sm.x_samples < 1 || sm.y_samples < 1 || (int64_t)sm.x_samples * sm.y_samples != sm.samples_per_pixel)) {
            g_err = "stratified sampler: samples_per_pixel must be x_samples * y_samples"; return MI_ERR_INVALID;
        }
        if (sm.type == MI_SAMPLER_ZEROTWO && (sm.samples_per_pixel & (sm.samples_per_pixel - 1)) != 0) { g_err = "02sequence sampler: samples_per_pixel must be a power of two"; return MI_ERR_INVALID; }
    }
    if (d->sampler.type == MI_SAMPLER_SOBOL) {
        if (!d->sampler.sobol_matrices || !d->sampler.sobol_vdc || !d->sampler.sobol_vdc_inv || d->sampler.sobol_log2_resolution < 0 ||
            d->sampler.sobol_log2_resolution > 25 || d->sampler.sobol_resolution != (1 << d->sampler.sobol_log2_resolution)) { g_err = "malformed Sobol' sampler tables"; return MI_ERR_INVALID; }
        if (d->sampler.n_sobol_dims < 6 + 8 * d->integrator.max_depth * d->integrator.n_ca_bands) {
            g_err = "Sobol' matrices cover too few dimensions for max_depth x n_ca_bands";
            return MI_ERR_INVALID;
        }
    }
    if (d->sampler.n_dims < (d->sampler.type == MI_SAMPLER_HALTON ? 6 + 8 * d->integrator.max_depth * d->integrator.n_ca_bands : 5)) {
        g_err = "Halton tables cover too few dimensions for max_depth x n_ca_bands (the reference's prime table ends at 1000)";
        return MI_ERR_INVALID;
    }

    HIPCHK(hipSetDevice(device_ordinal));
    mi_pt *pt = new mi_pt();
    pt->device = device_ordinal;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) pt->numCUs = prop.multiProcessorCount;
    }
    DScene &s = pt->scene;
    int rc;
#define UP(src, count, dst) if ((rc = Upload(pt, src, count, &dst)) != MI_OK) { mi_pt_destroy(pt); return rc; }
    {
        const float4 *nodes;
        UP((const float4 *)d->nodes, (size_t)d->n_nodes * 2, nodes);
        s.nodes = nodes;
    }
    for (uint32_t i = 0; i < d->n_nodes; ++i)
        if (d->nodes[i].n_prims > (unsigned)LEAF_COUNT_MASK) { g_err = "a BVH leaf holds more than 16383 primitives"; mi_pt_destroy(pt); return MI_ERR_UNSUPPORTED; }
    // a leaf's count as the traversal records carry it: | LEAF_SIMPLE when the leaf holds triangles only
    auto leafMeta = [&](const mi_bvh_node &leaf) -> unsigned {
        bool simple = getenv("MIPT_NO_COOP_LEAVES") == nullptr;
        for (uint32_t k = 0; k < leaf.n_prims && simple; ++k) {
            const mi_prim &p = d->prims[leaf.offset + k];
            simple = p.instance == 0 && p.shape >= 0;
        }
        return (unsigned)leaf.n_prims | (simple ? (unsigned)LEAF_SIMPLE : 0u);
    };
    s.bvhWidth = 4;   // MIPT_BVH_WIDTH=2: one record per BVH2 interior node, i.e. the reference's own node-visit counts
    if (const char *e = getenv("MIPT_BVH_WIDTH")) s.bvhWidth = (atoi(e) == 2) ? 2 : 4;
    if (s.bvhWidth == 4 && d->n_nodes > 0) {
        // Wide-4 records (OpenNode<4>): one per two-level subtree of the host's BVH2, in depth-first order. Slots 0/1 are the
        // left child's children (slot 0 alone = the left child itself when it is a leaf), slots 2/3 the right child's.
        const uint32_t nN = d->n_nodes;
        const mi_bvh_node *nodes = d->nodes;
        std::vector<int32_t> widx(nN, -1);
        std::vector<uint32_t> order;   // BVH2 roots of the records, in record order
        // the trees: the world's (root 0) and one per instanced object (mi_instance.root)
        std::vector<uint32_t> treeRoots(1, 0u);
        for (uint32_t k = 0; k < d->n_instances; ++k)
            if (std::find(treeRoots.begin(), treeRoots.end(), d->instances[k].root) == treeRoots.end()) treeRoots.push_back(d->instances[k].root);
        for (const uint32_t treeRoot : treeRoots) {
            if (nodes[treeRoot].n_prims > 0) {   // a single-leaf tree gets a record of its own: the leaf in slot 0
                widx[treeRoot] = (int32_t)order.size();
                order.push_back(treeRoot);
                continue;
            }
            std::vector<uint32_t> stack;
            stack.push_back(treeRoot);
            while (!stack.empty()) {
                const uint32_t i = stack.back();
                stack.pop_back();
                widx[i] = (int32_t)order.size();
                order.push_back(i);
                uint32_t sub[4];
                int nSub = 0;
                const uint32_t ch[2] = {i + 1, (uint32_t)nodes[i].offset};
                for (int c = 0; c < 2; ++c) {
                    if (nodes[ch[c]].n_prims > 0) continue;
                    const uint32_t g[2] = {ch[c] + 1, (uint32_t)nodes[ch[c]].offset};
                    for (int k = 0; k < 2; ++k) if (nodes[g[k]].n_prims == 0) sub[nSub++] = g[k];
                }
                for (int k = nSub - 1; k >= 0; --k) stack.push_back(sub[k]);   // first slot's subtree next in memory
            }
        }
        const size_t nWide = std::max<size_t>(order.size(), 1);
        std::vector<float4> w(nWide * 8, float4{0, 0, 0, 0});
        std::vector<int> need(nWide, 0);   // stack entries a ray can hold below this record (filled bottom-up)
        auto setBox = [&](float4 *rec, int slot, const mi_bvh_node &c) {
            float *f = (float *)rec + (slot < 2 ? 0 : 12) + (slot & 1) * 6;
            f[0] = c.bmin[0]; f[1] = c.bmin[1]; f[2] = c.bmin[2]; f[3] = c.bmax[0]; f[4] = c.bmax[1]; f[5] = c.bmax[2];
        };
        auto finish = [&](float4 *rec, const int link[4], const unsigned cnt[4], int axisRoot, const int groupN[2], const int groupAxis[2]) {
            memcpy(&rec[6], link, 16);
            const unsigned c01 = cnt[0] | (cnt[1] << 16), c23 = cnt[2] | (cnt[3] << 16);
            unsigned ord[2] = {0, 0};
            for (int oct = 0; oct < 8; ++oct) {
                auto neg = [&](int axis) { return ((oct >> axis) & 1) != 0; };
                int seq[4], n = 0;
                bool used[4] = {false, false, false, false};
                for (int pass = 0; pass < 2; ++pass) {
                    const int g = (neg(axisRoot) ? 1 : 0) ^ pass;   // near child's group first (bvh.cpp:686-692)
                    const int base = 2 * g;
                    if (groupN[g] == 2) {
                        const bool swap = neg(groupAxis[g]);
                        seq[n++] = base + (swap ? 1 : 0);
                        seq[n++] = base + (swap ? 0 : 1);
                    } else if (groupN[g] == 1) seq[n++] = base;
                }
                for (int k = 0; k < n; ++k) used[seq[k]] = true;
                for (int sl = 0; sl < 4 && n < 4; ++sl) if (!used[sl]) seq[n++] = sl;   // absent slots last (never hit)
                unsigned perm = 0;
                for (int k = 0; k < 4; ++k) perm |= (unsigned)seq[k] << (2 * k);
                ord[oct >> 2] |= perm << (8 * (oct & 3));
            }
            memcpy(&rec[7].x, &c01, 4); memcpy(&rec[7].y, &c23, 4);
            memcpy(&rec[7].z, &ord[0], 4); memcpy(&rec[7].w, &ord[1], 4);
        };
        {
            for (size_t r = order.size(); r-- > 0;) {   // records in reverse: a child record's stack need is known before its parent's
                const uint32_t i = order[r];
                float4 *rec = &w[r * 8];
                if (nodes[i].n_prims > 0) {   // single-leaf tree: the root itself in slot 0
                    int link1[4] = {nodes[i].offset, 0, 0, 0};
                    unsigned cnt1[4] = {leafMeta(nodes[i]), 0xffffu, 0xffffu, 0xffffu};
                    setBox(rec, 0, nodes[i]);
                    const int gN1[2] = {1, 0}, gA1[2] = {0, 0};
                    finish(rec, link1, cnt1, 0, gN1, gA1);
                    need[r] = 0;
                    continue;
                }
                int link[4] = {0, 0, 0, 0};
                unsigned cnt[4] = {0xffffu, 0xffffu, 0xffffu, 0xffffu};
                int gN[2] = {0, 0}, gA[2] = {0, 0}, below = 0, present = 0;
                const uint32_t ch[2] = {i + 1, (uint32_t)nodes[i].offset};
                for (int c = 0; c < 2; ++c) {
                    const mi_bvh_node &cn = nodes[ch[c]];
                    uint32_t sl[2];
                    if (cn.n_prims > 0) { gN[c] = 1; sl[0] = ch[c]; }
                    else { gN[c] = 2; gA[c] = cn.axis; sl[0] = ch[c] + 1; sl[1] = (uint32_t)cn.offset; }
                    for (int k = 0; k < gN[c]; ++k) {
                        const mi_bvh_node &gn = nodes[sl[k]];
                        const int slot = 2 * c + k;
                        setBox(rec, slot, gn);
                        if (gn.n_prims > 0) { link[slot] = gn.offset; cnt[slot] = leafMeta(gn); }
                        else { link[slot] = widx[sl[k]]; cnt[slot] = 0; below = std::max(below, need[widx[sl[k]]]); }
                        ++present;
                    }
                }
                need[r] = present - 1 + below;
                finish(rec, link, cnt, nodes[i].axis, gN, gA);
            }
        }
        int needAll = need[0];   // the world tree's need, plus -- inside an instance -- the return entry and the object tree's
        for (uint32_t k = 0; k < d->n_instances; ++k) needAll = std::max(needAll, need[0] + 1 + need[widx[d->instances[k].root]]);
        if (needAll > STACK_LDS + STACK_SPILL) {
            // (pbrt's own 64-entry stack bounds the BVH2 depth; a wide record can hold up to three entries per two levels)
            s.bvhWidth = 2;
        } else {
            const float4 *dev;
            UP(w.data(), w.size(), dev);
            s.wnodes = dev;
            std::vector<int32_t> instRoot(std::max<uint32_t>(d->n_instances, 1), 0);
            std::vector<float4> instBounds((size_t)std::max<uint32_t>(d->n_instances, 1) * 2, float4{0, 0, 0, 0});
            for (uint32_t k = 0; k < d->n_instances; ++k) {
                const mi_bvh_node &rn = nodes[d->instances[k].root];
                instRoot[k] = widx[d->instances[k].root];
                instBounds[2 * k] = float4{rn.bmin[0], rn.bmin[1], rn.bmin[2], 0};
                instBounds[2 * k + 1] = float4{rn.bmax[0], rn.bmax[1], rn.bmax[2], 0};
            }
            UP(instRoot.data(), instRoot.size(), s.instWideRoot);
            UP(instBounds.data(), instBounds.size(), s.instRootBounds);
        }
    }
    if (s.bvhWidth == 2 && d->n_instances > 0) {
        g_err = "object instances are traversed over the two-level BVH records (MIPT_BVH_WIDTH=4), which MIPT_BVH_WIDTH=2 or this scene's tree depth rules out";
        mi_pt_destroy(pt);
        return MI_ERR_UNSUPPORTED;
    }
    if (s.bvhWidth == 2) {   // wide-2 nodes: one 64-B record per interior node with both children's boxes
        const uint32_t nN = d->n_nodes;
        std::vector<int32_t> widx(nN, -1);
        uint32_t nInterior = 0;
        for (uint32_t i = 0; i < nN; ++i) if (d->nodes[i].n_prims == 0) widx[i] = (int32_t)nInterior++;
        const bool rootLeaf = nN > 0 && d->nodes[0].n_prims > 0;
        std::vector<float4> w((size_t)std::max<uint32_t>(nInterior, 1) * 4, float4{0, 0, 0, 0});
        auto putChild = [&](float4 *rec, int which, uint32_t child, int axis) {
            const mi_bvh_node &c = d->nodes[child];
            int link, meta;
            if (c.n_prims > 0) { link = c.offset; meta = (int)leafMeta(c); }
            else { link = widx[child]; meta = 0; }
            if (which == 0) {
                rec[0] = float4{c.bmin[0], c.bmin[1], c.bmin[2], c.bmax[0]};
                rec[1].x = c.bmax[1]; rec[1].y = c.bmax[2];
                meta |= axis << 16;
                memcpy(&rec[3].x, &link, 4); memcpy(&rec[3].z, &meta, 4);
            } else {
                rec[1].z = c.bmin[0]; rec[1].w = c.bmin[1];
                rec[2] = float4{c.bmin[2], c.bmax[0], c.bmax[1], c.bmax[2]};
                memcpy(&rec[3].y, &link, 4); memcpy(&rec[3].w, &meta, 4);
            }
        };
        if (rootLeaf) {  // single-leaf tree: the root itself is the left "child", no right child
            putChild(&w[0], 0, 0, 0);
            const int absent = 0xffff;
            memcpy(&w[3].w, &absent, 4);
        } else {
            for (uint32_t i = 0; i < nN; ++i) {
                if (d->nodes[i].n_prims != 0) continue;
                float4 *rec = &w[(size_t)widx[i] * 4];
                putChild(rec, 0, i + 1, d->nodes[i].axis);
                putChild(rec, 1, (uint32_t)d->nodes[i].offset, d->nodes[i].axis);
            }
        }
        const float4 *dev;
        UP(w.data(), w.size(), dev);
        s.wnodes = dev;
    }
    // shading classes: one per distinct lobe-type list, in order of first appearance
    std::vector<int> matClass(d->n_materials, 0);
    {
        std::vector<std::vector<int>> signatures;
        int classLobes[MAX_CLASSES] = {0};       // per class: the longest lobe list
        unsigned classTypes[MAX_CLASSES] = {0};  // per class: lobe types (bits 0..15) and fresnel kinds (bits 16..) present
        for (uint32_t i = 0; i < d->n_materials; ++i) {
            const mi_material &m = d->materials[i];
            std::vector<int> sig;
            for (int j = 0; j < m.n_bxdfs; ++j) { sig.push_back(m.bxdf[j].type); sig.push_back(m.bxdf[j].fresnel); }
            sig.push_back(m.textured ? 1 : 0);
            size_t c = 0;
            while (c < signatures.size() && signatures[c] != sig) ++c;
            if (c == signatures.size()) signatures.push_back(sig);
            matClass[i] = (int)std::min<size_t>(c, MISS_CLASS - 1);
            ((m.n_bxdfs > 2 || c >= (size_t)MISS_CLASS - 1) ? pt->largeClasses : pt->smallClasses) |= 1u << matClass[i];
            classLobes[matClass[i]] = std::max(classLobes[matClass[i]], (c >= (size_t)MISS_CLASS - 1) ? MI_MAX_BXDFS : (int)m.n_bxdfs);
            for (int j = 0; j < m.n_bxdfs; ++j) {
                classTypes[matClass[i]] |= (1u << m.bxdf[j].type) | (1u << (16 + m.bxdf[j].fresnel));
                if (m.bxdf[j].scaled) classTypes[matClass[i]] |= TM_SCALED;
            }
            if (m.textured) classTypes[matClass[i]] |= TM_TEXTURED;
        }
        pt->smallClasses &= ~pt->largeClasses;   // a shared overflow class runs the 8-lobe kernel
        pt->smallClasses |= 1u << MISS_CLASS;
        s.classMask = pt->smallClasses | pt->largeClasses;
        // kernels compiled for a subset of the lobe types take the classes that fit
        pt->diffuseClasses = 1u << MISS_CLASS;
        for (int c = 0; c < MISS_CLASS; ++c) {
            if (!((pt->smallClasses >> c) & 1u)) continue;
            if ((classTypes[c] & ~TM_DIFFUSE) == 0) pt->diffuseClasses |= 1u << c;
            else if ((classTypes[c] & ~TM_PLASTIC) == 0) pt->plasticClasses |= 1u << c;
        }
        for (int c = 0; c < MISS_CLASS; ++c)
            if (((pt->smallClasses >> c) & 1u) && !((pt->diffuseClasses | pt->plasticClasses) >> c & 1u) && !(classTypes[c] & TM_TEXTURED) &&
                (classTypes[c] & ~TM_GLASS) == 0)
                pt->glassClasses |= 1u << c;
        if (getenv("MIPT_NO_SPECIALISE")) pt->diffuseClasses = pt->plasticClasses = pt->glassClasses = 0;
        if (getenv("MIPT_ALL_LIGHTS")) pt->hasInfiniteLight = true;
        pt->smallClasses &= ~(pt->diffuseClasses | pt->plasticClasses | pt->glassClasses);
        unsigned disneyTextured = 0u;
        for (int c = 0; c < MISS_CLASS; ++c)
            if (classTypes[c] & TM_TEXTURED) {
                if (((pt->smallClasses >> c) & 1u) && !getenv("MIPT_NO_SPECIALISE")) {
                    if ((classTypes[c] & ~(TM_DIFFUSE | TM_TEXTURED)) == 0) { pt->texturedDiffuse |= 1u << c; pt->smallClasses &= ~(1u << c); }
                    else if ((classTypes[c] & ~(TM_PLASTIC | TM_TEXTURED)) == 0) { pt->texturedPlastic |= 1u << c; pt->smallClasses &= ~(1u << c); }
                }
                // (textured "disney" classes go to the eight-lobe instance whatever their lobe count: only that one reads the
                // rules that form their spectra from the colour -- LobeTexT::rules, d_bsdf.h)
                const unsigned disneyBits = (1u << MI_BXDF_DISNEY_DIFFUSE) | (1u << MI_BXDF_DISNEY_FAKE_SS) | (1u << MI_BXDF_DISNEY_RETRO) |
                                            (1u << MI_BXDF_DISNEY_SHEEN) | (1u << MI_BXDF_DISNEY_CLEARCOAT) | (1u << (16 + MI_FRESNEL_DISNEY));
                if (classTypes[c] & disneyBits) {
                    if ((pt->smallClasses >> c) & 1u) { pt->smallClasses &= ~(1u << c); pt->largeClasses |= 1u << c; }
                    disneyTextured |= 1u << c;
                }
                if ((pt->smallClasses >> c) & 1u) { pt->texturedSmall |= 1u << c; pt->smallClasses &= ~(1u << c); }
                if ((pt->largeClasses >> c) & 1u) { pt->texturedLarge |= 1u << c; pt->largeClasses &= ~(1u << c); }
            }
        if (!getenv("MIPT_NO_SPECIALISE"))
            for (int c = 0; c < MISS_CLASS; ++c) {   // the lobe sets of "uber" and "disney": instances without the rest of the BxDF code
                if (!((pt->largeClasses >> c) & 1u)) continue;
                if (classLobes[c] <= 4 && (classTypes[c] & ~TM_UBER) == 0) { pt->uberClasses |= 1u << c; pt->largeClasses &= ~(1u << c); }
                else if ((classTypes[c] & ~TM_DISNEY) == 0) { pt->disneyClasses |= 1u << c; pt->largeClasses &= ~(1u << c); }
            }
        if (!getenv("MIPT_NO_SPECIALISE"))
            for (int c = 0; c < MISS_CLASS; ++c)
                if (classLobes[c] <= 4) {
                    if ((pt->largeClasses >> c) & 1u) { pt->mediumClasses |= 1u << c; pt->largeClasses &= ~(1u << c); }
                    if (((pt->texturedLarge & ~disneyTextured) >> c) & 1u) { pt->texturedMedium |= 1u << c; pt->texturedLarge &= ~(1u << c); }
                }
    }
    // pre-gathered leaf records: positions of each BVH-ordered primitive + flags
    {
        std::vector<float4> pt3((size_t)d->n_prims * 3);
        for (uint32_t i = 0; i < d->n_prims; ++i) {
            const mi_prim &p = d->prims[i];
            float4 a{0, 0, 0, 0}, b{0, 0, 0, 0}, c{0, 0, 0, 0};
            unsigned flags = 0;
            int shapeIdx = 0;
            if (p.instance == 0 && p.shape >= 0) {
                const int32_t *v = &d->tri_indices[3 * p.shape];
                const float *P = d->P;
                a = float4{P[3 * v[0]], P[3 * v[0] + 1], P[3 * v[0] + 2], 0};
                b = float4{P[3 * v[1]], P[3 * v[1] + 1], P[3 * v[1] + 2], 0};
                c = float4{P[3 * v[2]], P[3 * v[2] + 1], P[3 * v[2] + 2], 0};
                shapeIdx = p.shape;
                // degenerate triangles are rejected by Triangle::Intersect (triangle.cpp:303-314):
                // decide it once here with the same arithmetic (doubles in Cross)
                auto cross = [](const float *u, const float *w, double *o) {
                    o[0] = (double)u[1] * w[2] - (double)u[2] * w[1];
                    o[1] = (double)u[2] * w[0] - (double)u[0] * w[2];
                    o[2] = (double)u[0] * w[1] - (double)u[1] * w[0];
                };
                const mi_mesh &m = d->meshes[d->tri_mesh[p.shape]];
                float uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};
                if (m.flags & MI_MESH_HAS_UV) for (int k = 0; k < 3; ++k) { uv[k][0] = d->UV[2 * v[k]]; uv[k][1] = d->UV[2 * v[k] + 1]; }
                float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]}, duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
                float dp02[3] = {a.x - c.x, a.y - c.y, a.z - c.z}, dp12[3] = {b.x - c.x, b.y - c.y, b.z - c.z};
                float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
                bool degenerateUV = std::abs(determinant) < 1e-8;
                bool needNg = degenerateUV;
                if (!degenerateUV) {
                    float invdet = 1 / determinant;
                    float dpdu[3], dpdv[3];
                    for (int k = 0; k < 3; ++k) {
                        dpdu[k] = (duv12[1] * dp02[k] - duv02[1] * dp12[k]) * invdet;
                        dpdv[k] = (-duv12[0] * dp02[k] + duv02[0] * dp12[k]) * invdet;
                    }
                    double cr[3];
                    cross(dpdu, dpdv, cr);
                    float cx = (float)cr[0], cy = (float)cr[1], cz = (float)cr[2];
                    if (cx * cx + cy * cy + cz * cz == 0) needNg = true;
                }
                if (needNg) {
                    float e1[3] = {c.x - a.x, c.y - a.y, c.z - a.z}, e2[3] = {b.x - a.x, b.y - a.y, b.z - a.z};
                    double cr[3];
                    cross(e1, e2, cr);
                    float cx = (float)cr[0], cy = (float)cr[1], cz = (float)cr[2];
                    if (cx * cx + cy * cy + cz * cz == 0) flags |= PRIM_FLAG_DEGENERATE;
                }
                if (m.alpha_tex >= 0 || m.shadow_alpha_tex >= 0) { flags |= PRIM_FLAG_ALPHA; pt->hasAlphaMasks = true; }
            } else if (p.instance == 0) {
                pt->hasQuadrics = true;
                flags |= PRIM_FLAG_SPHERE;
                shapeIdx = ~p.shape;
            }
            if (p.instance != 0) {   // a TransformedPrimitive: no shape of its own
                a = b = c = float4{0, 0, 0, 0};
                flags = PRIM_FLAG_INSTANCE;
                shapeIdx = p.instance - 1;
            }
            flags |= (unsigned)(p.material >= 0 ? matClass[p.material] : MISS_CLASS) << PRIM_CLASS_SHIFT;
            memcpy(&a.w, &flags, 4);
            memcpy(&b.w, &shapeIdx, 4);
            pt3[3 * i] = a; pt3[3 * i + 1] = b; pt3[3 * i + 2] = c;
        }
        const float4 *dev;
        UP(pt3.data(), pt3.size(), dev);
        s.primTri = dev;
    }
    UP(d->prims, d->n_prims, s.prims);
    s.nInstances = d->n_instances;
    s.instances = nullptr;
    if (d->n_instances) UP(d->instances, d->n_instances, s.instances);
    pt->hasInstances = d->n_instances > 0;
    UP(d->tri_indices, (size_t)d->n_tris * 3, s.triIndices);
    UP(d->tri_mesh, d->n_tris, s.triMesh);
    UP(d->P, (size_t)d->n_verts * 3, s.P);
    UP(d->N, (size_t)d->n_verts * 3, s.N);
    UP(d->UV, (size_t)d->n_verts * 2, s.UV);
    UP(d->meshes, d->n_meshes, s.meshes);
    UP(d->spheres, d->n_spheres, s.spheres);
    UP(d->materials, d->n_materials, s.materials);
    UP(d->lights, d->n_lights, s.lights);
    {   // dilated world bounds of the area lights' shapes (F_MIS_DARK)
        std::vector<float4> lb((size_t)std::max<uint32_t>(d->n_lights, 1) * 2, float4{-INFINITY, -INFINITY, -INFINITY, 0});
        for (uint32_t i = 0; i < d->n_lights; ++i) {
            const mi_light &l = d->lights[i];
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            auto add = [&](float x, float y, float z) { const float p[3] = {x, y, z}; for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } };
            bool have = false;
            if (l.type == MI_LIGHT_DIFFUSE_AREA && l.shape >= 0 && (uint32_t)l.shape < d->n_tris) {
                const int32_t *v = &d->tri_indices[3 * l.shape];
                for (int k = 0; k < 3; ++k) add(d->P[3 * v[k]], d->P[3 * v[k] + 1], d->P[3 * v[k] + 2]);
                have = true;
            } else if (l.type == MI_LIGHT_DIFFUSE_AREA && l.shape < 0 && (uint32_t)(~l.shape) < d->n_spheres) {
                const mi_sphere &sp = d->spheres[~l.shape];   // Sphere::ObjectBound through ObjectToWorld (shape.cpp:50)
                for (int c = 0; c < 8; ++c) {
                    const float x = (c & 1) ? sp.radius : -sp.radius, y = (c & 2) ? sp.radius : -sp.radius, z = (c & 4) ? sp.z_max : sp.z_min;
                    const float *m = sp.o2w;
                    const float w = m[12] * x + m[13] * y + m[14] * z + m[15];
                    add((m[0] * x + m[1] * y + m[2] * z + m[3]) / w, (m[4] * x + m[5] * y + m[6] * z + m[7]) / w, (m[8] * x + m[9] * y + m[10] * z + m[11]) / w);
                }
                have = true;
            }
            if (have) {
                float scale = 0;
                for (int a = 0; a < 3; ++a) scale = std::max(scale, std::max(std::abs(mn[a]), std::abs(mx[a])) + (mx[a] - mn[a]));
                const float e = 1e-3f * scale;
                lb[2 * i] = float4{mn[0] - e, mn[1] - e, mn[2] - e, 0};
                lb[2 * i + 1] = float4{mx[0] + e, mx[1] + e, mx[2] + e, 0};
            } else {   // not an area light: everything may hit
                lb[2 * i] = float4{-INFINITY, -INFINITY, -INFINITY, 0};
                lb[2 * i + 1] = float4{INFINITY, INFINITY, INFINITY, 0};
            }
        }
        UP(lb.data(), lb.size(), s.lightBounds);
    }
    {   // the area lights' primitives, and whether the MIS rays can be asked as visibility queries (k_trav, MODE 3): no
        // instances (the traversal kernels compiled for those keep the closest-hit form), no alpha mask on an emitter's own
        // mesh, and every area light the shape of exactly one primitive
        std::vector<int> lp((size_t)std::max<uint32_t>(d->n_lights, 1), (int)MIS_EXCL_NONE), seen((size_t)std::max<uint32_t>(d->n_lights, 1), 0);
        bool ok = !pt->hasInstances && d->n_prims < MIS_EXCL_NONE;
        for (uint32_t i = 0; i < d->n_prims; ++i) {
            const int al = d->prims[i].area_light;
            if (al < 0) continue;
            if ((uint32_t)al >= d->n_lights) { ok = false; continue; }
            lp[al] = (int)i;
            if (++seen[al] > 1 || d->prims[i].shape != d->lights[al].shape || d->prims[i].instance != 0) ok = false;
        }
        for (uint32_t i = 0; i < d->n_lights; ++i) {
            const mi_light &l = d->lights[i];
            if (l.type != MI_LIGHT_DIFFUSE_AREA) continue;
            if (!seen[i]) ok = false;
            // an emitter whose own mesh is masked: Shape::Pdf intersects it without the mask (shape.cpp:60), the traversal with it
            if (l.shape >= 0 && (uint32_t)l.shape < d->n_tris) {
                const mi_mesh &m = d->meshes[d->tri_mesh[l.shape]];
                if (m.alpha_tex >= 0 || m.shadow_alpha_tex >= 0) ok = false;
            }
        }
        UP(lp.data(), lp.size(), s.lightPrim);
#ifdef MIPT_NO_MIS_ANY
        ok = false;
#endif
        s.misAny = ok ? 1 : 0;
    }
    UP(d->sampler.primes, d->sampler.n_dims, s.primes);
    UP(d->sampler.prime_sums, d->sampler.n_dims, s.primeSums);
    UP(d->sampler.perms, d->sampler.n_perms, s.perms);
    UP(d->film.filter_table, 256, s.filterTable);
    s.sobolMatrices = nullptr; s.sobolVdc = nullptr; s.sobolVdcInv = nullptr;
    if (d->sampler.type == MI_SAMPLER_SOBOL) {
        UP(d->sampler.sobol_matrices, (size_t)d->sampler.n_sobol_dims * MI_SOBOL_MATRIX_SIZE, s.sobolMatrices);
        UP(d->sampler.sobol_vdc, (size_t)MI_SOBOL_MATRIX_SIZE, s.sobolVdc);
        UP(d->sampler.sobol_vdc_inv, (size_t)MI_SOBOL_MATRIX_SIZE, s.sobolVdcInv);
    }
    {   // division magics and the per-pixel Halton offsets (GetIndexForSample, halton.cpp:98-118)
        std::vector<uint64_t> magic(d->sampler.n_dims);
        for (int i = 0; i < d->sampler.n_dims; ++i) {
            const uint64_t p = (uint64_t)d->sampler.primes[i];
            magic[i] = (~0ull) / p + 1;  // ceil(2^64 / p) for p not a power of two; p == 2 is never divided here
        }
        UP(magic.data(), magic.size(), s.primeMagic);
        std::vector<uint32_t> table(128 * 128, 0);
        const mi_sampler &sm = d->sampler;
        if (sm.sample_stride > 1) {
            auto inverseRadicalInverse = [](uint64_t base, uint64_t inverse, int nDigits) {
                uint64_t index = 0;
                for (int i = 0; i < nDigits; ++i) { uint64_t digit = inverse % base; inverse /= base; index = index * base + digit; }
                return index;
            };
            for (int py = 0; py < 128; ++py)
                for (int px = 0; px < 128; ++px) {
                    int64_t offset = 0;
                    const int pm[2] = {px, py};
                    for (int i = 0; i < 2; ++i) {
                        uint64_t dimOffset = inverseRadicalInverse(i == 0 ? 2 : 3, (uint64_t)pm[i], sm.base_exponents[i]);
                        offset += (int64_t)(dimOffset * (uint64_t)(sm.sample_stride / sm.base_scales[i]) * (uint64_t)sm.mult_inverse[i]);
                    }
                    offset %= (int64_t)sm.sample_stride;
                    table[py * 128 + px] = (uint32_t)offset;
                }
        }
        UP(table.data(), table.size(), s.pixelOffsetTable);
    }
    s.nNodes = d->n_nodes; s.nPrims = d->n_prims; s.nLights = d->n_lights; s.nMaterials = d->n_materials;
    for (int i = 0; i < MI_NSPEC; ++i) s.cieY[i] = d->cie_y[i];
    {   // environment maps of the infinite lights: tables to the device, then the records that point at them
        std::vector<mi_envmap> envs(d->n_envmaps);
        for (uint32_t i = 0; i < d->n_envmaps; ++i) {
            const mi_envmap &e = d->envmaps[i];
            if (e.width < 1 || e.height < 1 || e.nu < 1 || e.nv < 1 || !e.rgb || !e.cond_func || !e.cond_cdf || !e.cond_func_int ||
                !e.marg_func || !e.marg_cdf) { g_err = "malformed mi_envmap"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
            mi_envmap m = e;
            UP(e.rgb, (size_t)e.width * e.height * 3, m.rgb);
            UP(e.cond_func, (size_t)e.nu * e.nv, m.cond_func);
            UP(e.cond_cdf, (size_t)(e.nu + 1) * e.nv, m.cond_cdf);
            UP(e.cond_func_int, (size_t)e.nv, m.cond_func_int);
            UP(e.marg_func, (size_t)e.nv, m.marg_func);
            UP(e.marg_cdf, (size_t)e.nv + 1, m.marg_cdf);
            envs[i] = m;
        }
        s.envmaps = nullptr;
        if (!envs.empty()) UP(envs.data(), envs.size(), s.envmaps);
        UP(&d->rgb_illum[0][0], (size_t)7 * MI_NSPEC, s.rgbIllum);
        {   // image textures: pyramids into HBM, mi_mipmap records with device texel pointers
            std::vector<mi_mipmap> mips(d->n_mipmaps);
            for (uint32_t i = 0; i < d->n_mipmaps; ++i) {
                mi_mipmap m = d->mipmaps[i];
                if (m.n_levels < 1 || m.n_levels > MI_MAX_MIP_LEVELS || !m.texels || m.width < 1 || m.height < 1) { g_err = "malformed mi_mipmap"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
                size_t nTexels = 0;
                for (int l = 0; l < m.n_levels; ++l) nTexels = std::max<size_t>(nTexels, (size_t)m.level_offset[l] + (size_t)std::max(1, m.width >> l) * std::max(1, m.height >> l));
                UP(d->mipmaps[i].texels, nTexels * 3, m.texels);
                mips[i] = m;
            }
            s.mipmaps = nullptr; s.textures = nullptr;
            if (!mips.empty()) UP(mips.data(), mips.size(), s.mipmaps);
            for (uint32_t i = 0; i < d->n_textures; ++i)
                if (d->textures[i].type == MI_TEX_IMAGEMAP && (uint32_t)d->textures[i].mipmap >= d->n_mipmaps) { g_err = "mi_texture.mipmap out of range"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
            if (d->n_textures) UP(d->textures, (size_t)d->n_textures, s.textures);
            pt->nTextures = d->n_textures;
            for (uint32_t i = 0; i < d->n_textures; ++i) pt->textureTypes.push_back(d->textures[i].type);
            float lut[128];   // MIPMap::weightLut, mipmap.h:199-206
            for (int i = 0; i < 128; ++i) {
                float alpha = 2;
                float r2 = float(i) / float(128 - 1);
                lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
            }
            UP(lut, (size_t)128, s.ewaWeights);
            s.invSqrtSpp = 1 / std::sqrt((float)d->sampler.samples_per_pixel);
            for (uint32_t i = 0; i < d->n_meshes; ++i)
                if (d->meshes[i].alpha_tex >= (int)d->n_textures || d->meshes[i].shadow_alpha_tex >= (int)d->n_textures) { g_err = "mi_mesh alpha texture index out of range"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
            for (uint32_t i = 0; i < d->n_materials; ++i)
                for (int j = 0; d->materials[i].textured && j < d->materials[i].n_bxdfs; ++j) {
                    const mi_lobe_tex &t = d->materials[i].tex[j];
                    if (t.tex_R >= (int)d->n_textures || t.tex_S >= (int)d->n_textures) { g_err = "mi_lobe_tex texture index out of range"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
                }
        }
        s.nInfiniteLights = 0;
        for (int k = 0; k < 4; ++k) s.infiniteLights[k] = -1;
        for (uint32_t i = 0; i < d->n_lights; ++i)
            if (d->lights[i].type == MI_LIGHT_INFINITE) {
                if ((uint32_t)d->lights[i].envmap >= d->n_envmaps) { g_err = "infinite light without environment map"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
                if (s.nInfiniteLights == 4) { g_err = "more than 4 infinite lights"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
                s.infiniteLights[s.nInfiniteLights++] = (int)i;
                pt->hasInfiniteLight = true;
            }
    }
    s.camera = d->camera;
    for (int i = 0; i < 4; ++i) { s.croppedBounds[i] = d->film.cropped_bounds[i]; s.sampleBounds[i] = d->film.sample_bounds[i]; s.pixelBounds[i] = d->integrator.pixel_bounds[i]; }
    s.filterRadius[0] = d->film.filter_radius[0]; s.filterRadius[1] = d->film.filter_radius[1];
    s.maxSampleLuminance = d->film.max_sample_luminance;
    for (int i = 0; i < 2; ++i) { s.baseScales[i] = d->sampler.base_scales[i]; s.baseExponents[i] = d->sampler.base_exponents[i]; s.multInverse[i] = d->sampler.mult_inverse[i]; }
    s.sampleStride = d->sampler.sample_stride;
    s.sampleAtPixelCenter = d->sampler.sample_at_pixel_center;
    s.samplerType = d->sampler.type;
    s.samplesPerPixel = d->sampler.samples_per_pixel;
    s.pixTab1 = s.pixTab2 = nullptr;
    s.index32 = 0; s.storePixelSample = 1;
    s.pixelDims = d->sampler.pixel_dims; s.xSamples = d->sampler.x_samples; s.ySamples = d->sampler.y_samples; s.jitter = d->sampler.jitter;
    s.sobolResolution = d->sampler.sobol_resolution;
    s.sobolLog2Resolution = d->sampler.sobol_log2_resolution;
    s.maxDepth = d->integrator.max_depth;
    s.rrThreshold = d->integrator.rr_threshold;
    s.nBands = d->integrator.n_ca_bands;
    s.bandDelta = (int)std::round((float)MI_NSPEC / (float)s.nBands);  // spectralpath.cpp:258
    pt->spp = d->sampler.samples_per_pixel;
    if (d->n_nodes) for (int i = 0; i < 3; ++i) { s.wbMin[i] = d->nodes[0].bmin[i]; s.wbMax[i] = d->nodes[0].bmax[i]; }
    if (d->sampler.type >= MI_SAMPLER_ZEROTWO && d->sampler.pixel_dims > 0) {
        // the pixel samplers' tables: every sampled dimension of every sample of every pixel of the sample bounds (StartPixel of
        // the reference, done once for all pixels), 12 bytes per (pixel, dimension, sample)
        const size_t nPix = (size_t)(s.sampleBounds[2] - s.sampleBounds[0]) * (size_t)(s.sampleBounds[3] - s.sampleBounds[1]);
        const size_t nVal = nPix * (size_t)s.pixelDims * (size_t)s.samplesPerPixel;
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); freeB = ~(size_t)0; }
        if (nVal * 12 > freeB / 2) { g_err = "the pixel sampler's tables (" + std::to_string(nVal * 12 >> 20) + " MiB) exceed half the free device memory"; mi_pt_destroy(pt); return MI_ERR_NOMEM; }
        float *t1 = nullptr, *t2 = nullptr;
        if (hipMalloc((void **)&t1, nVal * 4 + 16) != hipSuccess || hipMalloc((void **)&t2, nVal * 8 + 16) != hipSuccess) {
            (void)hipGetLastError();
            if (t1) hipFree(t1);
            g_err = "hipMalloc(pixel sampler tables) failed"; mi_pt_destroy(pt); return MI_ERR_NOMEM;
        }
        pt->allocs.push_back(t1); pt->allocs.push_back(t2);
        s.pixTab1 = t1; s.pixTab2 = t2;
        hipLaunchKernelGGL(k_pixel_tables, dim3((unsigned)((nPix + 127) / 128)), dim3(128), 0, 0, s, t1, t2, (unsigned long long)nPix);
        if (hipDeviceSynchronize() != hipSuccess) { g_err = "k_pixel_tables failed"; mi_pt_destroy(pt); return MI_ERR_HIP; }
    }
    // light-selection distributions
    s.ldType = d->light_distrib.type;
    for (int i = 0; i < 3; ++i) s.nVoxels[i] = d->light_distrib.n_voxels[i];
    if (d->n_lights == 0) {
        UP((const float *)nullptr, 0, s.ldFunc); UP((const float *)nullptr, 0, s.ldCdf); UP((const float *)nullptr, 0, s.ldFuncInt);
    } else if (s.ldType != MI_LD_SPATIAL) {
        if (!d->light_distrib.func || !d->light_distrib.cdf || !d->light_distrib.func_int) { g_err = "light distribution tables missing"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
        UP(d->light_distrib.func, d->n_lights, s.ldFunc);
        UP(d->light_distrib.cdf, d->n_lights + 1, s.ldCdf);
        UP(d->light_distrib.func_int, 1, s.ldFuncInt);
    } else {
        size_t nVox = (size_t)s.nVoxels[0] * s.nVoxels[1] * s.nVoxels[2];
        if (nVox == 0 || nVox * d->n_lights > (1ull << 31)) { g_err = "spatial light distribution too large for the dense per-voxel table"; mi_pt_destroy(pt); return MI_ERR_UNSUPPORTED; }
        float *f, *c, *fi;
        if (hipMalloc((void **)&f, nVox * d->n_lights * 4) != hipSuccess || hipMalloc((void **)&c, nVox * (d->n_lights + 1) * 4) != hipSuccess ||
            hipMalloc((void **)&fi, nVox * 4) != hipSuccess) { g_err = "hipMalloc(light distribution) failed"; mi_pt_destroy(pt); return MI_ERR_NOMEM; }
        pt->allocs.push_back(f); pt->allocs.push_back(c); pt->allocs.push_back(fi);
        s.ldFunc = f; s.ldCdf = c; s.ldFuncInt = fi;
        hipLaunchKernelGGL(k_build_spatial, dim3((unsigned)((nVox + 127) / 128)), dim3(128), 0, 0, s, f, c, fi, (uint32_t)nVox);
        if (hipDeviceSynchronize() != hipSuccess) { g_err = "k_build_spatial failed"; mi_pt_destroy(pt); return MI_ERR_HIP; }
    }
#undef UP
    pt->filmW = d->film.cropped_bounds[2] - d->film.cropped_bounds[0];
    pt->filmH = d->film.cropped_bounds[3] - d->film.cropped_bounds[1];
    if (pt->filmW <= 0 || pt->filmH <= 0) { g_err = "empty film"; mi_pt_destroy(pt); return MI_ERR_INVALID; }
    pt->nPix = (size_t)pt->filmW * pt->filmH;
    if (hipMalloc((void **)&pt->film, pt->nPix * 32 * sizeof(float)) != hipSuccess) { g_err = "hipMalloc(film) failed"; mi_pt_destroy(pt); return MI_ERR_NOMEM; }
    hipMemset(pt->film, 0, pt->nPix * 32 * sizeof(float));
    {
        int nSub = 1;  // sub-renderers running concurrently on their own streams (MIPT_STREAMS overrides, 1..8)
        if (const char *e = getenv("MIPT_STREAMS")) nSub = std::max(1, std::min(8, atoi(e)));
        pt->subs.resize(nSub);
        for (SubRenderer &sub : pt->subs) {
            if (hipMalloc((void **)&sub.ctr, sizeof(DevCounters)) != hipSuccess) { g_err = "hipMalloc(counters) failed"; mi_pt_destroy(pt); return MI_ERR_NOMEM; }
            if (hipStreamCreateWithFlags(&sub.stream, hipStreamNonBlocking) != hipSuccess) { g_err = "hipStreamCreate failed"; mi_pt_destroy(pt); return MI_ERR_HIP; }
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < N_EV; ++b)
                    if (hipEventCreate(&sub.evIter[a][b]) != hipSuccess) { g_err = "hipEventCreate failed"; mi_pt_destroy(pt); return MI_ERR_HIP; }
        }
    }
    *out = pt;
    return MI_OK;
}

#ifdef MIPT_SORT_EXPERIMENT
// Experiment (not part of the product build): how much would the traversal kernels gain from coherent work lists? Sorts a
// ray queue by (origin cell 8x8x8, direction octant) with hipcub before the traversal launch; MIPT_SORT = bit mask of the
// modes to sort (1: continuation rays, 2: shadow rays, 4: MIS rays). The sort's own time is not the question here: the
// traversal kernels' durations in a rocprofv3 kernel trace are.
__global__ void k_sort_keys(DScene s, Pool pool, const uint32_t *queue, unsigned n, int mode, uint32_t *keys) {
    const unsigned i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = queue[i];
    const float4 r0 = pool.R(mode == 0 ? R_RAY0 : (mode == 1 ? R_SH0 : R_MI0), slot), r1 = pool.R(mode == 0 ? R_RAY1 : (mode == 1 ? R_SH1 : R_MI1), slot);
    const float dx = mode == 0 ? r1.x : r0.w, dy = mode == 0 ? r1.y : r1.x, dz = mode == 0 ? r1.z : r1.y;
    auto cell = [&](float v, int a) { const float t = (v - s.wbMin[a]) / (s.wbMax[a] - s.wbMin[a]); return (unsigned)min(7, max(0, (int)(t * 8.f))); };
    const unsigned cx = cell(r0.x, 0), cy = cell(r0.y, 1), cz = cell(r0.z, 2);
    unsigned m = 0;
    for (int b = 0; b < 3; ++b) m |= (((cx >> b) & 1u) << (3 * b)) | (((cy >> b) & 1u) << (3 * b + 1)) | (((cz >> b) & 1u) << (3 * b + 2));
    keys[i] = (m << 3) | (dx < 0 ? 1u : 0u) | (dy < 0 ? 2u : 0u) | (dz < 0 ? 4u : 0u);
}
static void SortQueueExperiment(mi_pt *pt, SubRenderer &sub, int mode) {
    static int mask = getenv("MIPT_SORT") ? atoi(getenv("MIPT_SORT")) : 0;
    if (!((mask >> mode) & 1)) return;
    hipStream_t st = sub.stream;
    static uint32_t *keys = nullptr, *keysOut = nullptr, *valsOut = nullptr;
    static void *temp = nullptr;
    static size_t tempBytes = 0;
    const size_t cap = sub.pool.n;
    if (!keys) {
        hipMalloc((void **)&keys, cap * 4); hipMalloc((void **)&keysOut, cap * 4); hipMalloc((void **)&valsOut, cap * 4);
        hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, keys, keysOut, valsOut, valsOut, (int)cap, 0, 12, st);
        hipMalloc(&temp, tempBytes);
    }
    unsigned cnt[2] = {0, 0};
    uint32_t *queue;
    if (mode == 0) { hipMemcpyAsync(&cnt[0], &sub.ctr->contCount.v, 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); queue = sub.pool.extQ + (sub.pool.n - cnt[0]); }
    else { hipMemcpyAsync(&cnt[0], mode == 1 ? &sub.ctr->shadowCount.v : &sub.ctr->misCount.v, 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); queue = mode == 1 ? sub.pool.shadowQ : sub.pool.misQ; }
    const unsigned n = cnt[0];
    if (n < 2) return;
    hipLaunchKernelGGL(k_sort_keys, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, pt->scene, sub.pool, queue, n, mode, keys);
    size_t tb = tempBytes;
    hipcub::DeviceRadixSort::SortPairs(temp, tb, keys, keysOut, queue, valsOut, (int)n, 0, 12, st);
    hipMemcpyAsync(queue, valsOut, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
}
#endif

// The launches of one wavefront iteration, shared by RenderSub and the path-dump tool.
static void LaunchTraversal(mi_pt *pt, SubRenderer &sub, int mode, dim3 travGrid, bool closestMis = false) {
    if (mode == 2 && pt->scene.misAny && !closestMis) mode = 3;   // the MIS rays as visibility queries (k_trav, MODE 3)
    if (TRAV_IS_ANY(mode) && !pt->hasAlphaMasks && !pt->hasInstances && travGrid.x == (unsigned)pt->numCUs * TRAV_BLOCKS_PER_CU) travGrid.x = (unsigned)pt->numCUs * MIPT_TRAV_WAVES_PER_EU_ANY;   // (a full-size launch: one more block per CU)
    const DScene &s = pt->scene;
    const dim3 block(BLOCK);
    hipStream_t st = sub.stream;
#define TRAV_LAUNCH(MODE_, ALPHA_, W_) hipLaunchKernelGGL((k_trav<MODE_, ALPHA_, W_>), travGrid, block, 0, st, s, sub.pool, sub.ctr)
#define TRAV_LAUNCH_W(MODE_, ALPHA_) do { if (s.bvhWidth == 4) TRAV_LAUNCH(MODE_, ALPHA_, 4); else TRAV_LAUNCH(MODE_, ALPHA_, 2); } while (0)
    if (pt->hasInstances) {   // (mi_pt_create: instanced scenes always have the two-level records)
        if (mode == 0) hipLaunchKernelGGL((k_trav<0, true, 4, true>), travGrid, block, 0, st, s, sub.pool, sub.ctr);
        else if (mode == 1) hipLaunchKernelGGL((k_trav<1, true, 4, true>), travGrid, block, 0, st, s, sub.pool, sub.ctr);
        else hipLaunchKernelGGL((k_trav<2, true, 4, true>), travGrid, block, 0, st, s, sub.pool, sub.ctr);
    } else if (pt->hasAlphaMasks) {
        if (mode == 0) TRAV_LAUNCH_W(0, true);
        else if (mode == 1) TRAV_LAUNCH_W(1, true);
        else if (mode == 3) TRAV_LAUNCH_W(3, true);
        else TRAV_LAUNCH_W(2, true);
    } else {
        if (mode == 0) TRAV_LAUNCH_W(0, false);
        else if (mode == 1) TRAV_LAUNCH_W(1, false);
        else if (mode == 3) TRAV_LAUNCH_W(3, false);
        else TRAV_LAUNCH_W(2, false);
    }
#undef TRAV_LAUNCH_W
#undef TRAV_LAUNCH
}

static void LaunchShade(mi_pt *pt, SubRenderer &sub, dim3 grid) {
    const DScene &s = pt->scene;
    const dim3 block(BLOCK);
    hipStream_t st = sub.stream;
    const dim3 shadeGrid(grid.x + MAX_CLASSES);
#ifdef MIPT_HOT_ONLY
    if (pt->diffuseClasses) hipLaunchKernelGGL((k_shade<2, TM_DIFFUSE | TM_LIGHTS_NO_ENV>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->diffuseClasses);
    if (pt->plasticClasses) hipLaunchKernelGGL((k_shade<2, TM_PLASTIC | TM_LIGHTS_NO_ENV>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->plasticClasses);
    return;
#else
    if (pt->hasInstances) {   // scenes with object instances: the two fully general instances of the kernel, by lobe count
        const unsigned two = pt->diffuseClasses | pt->plasticClasses | pt->glassClasses | pt->smallClasses | pt->texturedDiffuse | pt->texturedPlastic | pt->texturedSmall;
        const unsigned more = pt->mediumClasses | pt->texturedMedium | pt->largeClasses | pt->texturedLarge | pt->uberClasses | pt->disneyClasses;
        if (two) hipLaunchKernelGGL((k_shade<2, TM_ALL>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, two);
        if (more) hipLaunchKernelGGL((k_shade<MI_MAX_BXDFS, TM_ALL>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, more);
        return;
    }
    // the two hot instances (matte-like and plastic-like classes) exist with and without environment-light code and
    // with the Halton sampler alone or all three
#define SHADE_LAUNCH(TM_, CLASSES_) hipLaunchKernelGGL((k_shade<2, TM_>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, CLASSES_)
#define SHADE_LAUNCH_HOT(TM_, CLASSES_) do { if (!(CLASSES_)) break; \
        if (halton) { if (pt->hasInfiniteLight) SHADE_LAUNCH(TM_ | TM_LIGHTS_ALL, CLASSES_); else SHADE_LAUNCH(TM_ | TM_LIGHTS_NO_ENV, CLASSES_); } \
        else { if (pt->hasInfiniteLight) SHADE_LAUNCH(TM_ | TM_LIGHTS_ALL | TM_SAMPLERS, CLASSES_); else SHADE_LAUNCH(TM_ | TM_LIGHTS_NO_ENV | TM_SAMPLERS, CLASSES_); } } while (0)
    const bool halton = s.samplerType == MI_SAMPLER_HALTON;
    SHADE_LAUNCH_HOT(TM_DIFFUSE, pt->diffuseClasses);
    SHADE_LAUNCH_HOT(TM_PLASTIC, pt->plasticClasses);
#undef SHADE_LAUNCH_HOT
#undef SHADE_LAUNCH
    if (pt->smallClasses) hipLaunchKernelGGL((k_shade<2, TM_GENERIC>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->smallClasses);
    if (pt->glassClasses) hipLaunchKernelGGL((k_shade<2, TM_GLASS | TM_LIGHTS_ALL | TM_SAMPLERS>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->glassClasses);
    if (pt->uberClasses) hipLaunchKernelGGL((k_shade<4, TM_UBER | TM_LIGHTS_ALL | TM_SAMPLERS>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->uberClasses);
    if (pt->disneyClasses) hipLaunchKernelGGL((k_shade<MI_MAX_BXDFS, TM_DISNEY | TM_LIGHTS_ALL | TM_SAMPLERS>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->disneyClasses);
    if (pt->mediumClasses) hipLaunchKernelGGL((k_shade<4, TM_GENERIC>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->mediumClasses);
    if (pt->texturedMedium) hipLaunchKernelGGL((k_shade<4, TM_FULL>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->texturedMedium);
    if (pt->largeClasses) hipLaunchKernelGGL((k_shade<MI_MAX_BXDFS, TM_GENERIC>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->largeClasses);
    if (pt->texturedDiffuse) hipLaunchKernelGGL((k_shade<2, TM_DIFFUSE | TM_TEXTURED | TM_LIGHTS_ALL | TM_SAMPLERS>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->texturedDiffuse);
    if (pt->texturedPlastic) hipLaunchKernelGGL((k_shade<2, TM_PLASTIC | TM_TEXTURED | TM_LIGHTS_ALL | TM_SAMPLERS>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->texturedPlastic);
    if (pt->texturedSmall) hipLaunchKernelGGL((k_shade<2, TM_FULL>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->texturedSmall);
    if (pt->texturedLarge) hipLaunchKernelGGL((k_shade<MI_MAX_BXDFS, TM_FULL>), shadeGrid, block, 0, st, s, sub.pool, sub.ctr, pt->texturedLarge);
#endif   // MIPT_HOT_ONLY
}

// One sub-renderer = one path pool with its queues and counters on its own HIP stream.
// mi_pt_render splits its tile shard over pt->subs.size() sub-renderers that run
// concurrently (one host thread each): while one pool is in a traversal kernel's tail
// the other pool's kernels fill the idle CUs, which a single dependent chain of launches
// cannot do.
static int RenderSub(mi_pt *pt, SubRenderer &sub, const mi_render_params *rp, int subIndex, int subCount) {
    HIPCHK(hipSetDevice(pt->device));
    hipStream_t st = sub.stream;
    const DScene &s = pt->scene;
    WorkDesc wd{};
    wd.nTilesX = (s.sampleBounds[2] - s.sampleBounds[0] + 15) / 16;
    wd.nTilesY = (s.sampleBounds[3] - s.sampleBounds[1] + 15) / 16;
    const int nTiles = wd.nTilesX * wd.nTilesY;
    wd.shardIndex = rp->shard_index + rp->shard_count * subIndex;
    wd.shardCount = rp->shard_count * subCount;
    wd.nTilesShard = (wd.shardIndex < nTiles) ? (nTiles - wd.shardIndex + wd.shardCount - 1) / wd.shardCount : 0;
    wd.spp = rp->spp_override > 0 ? rp->spp_override : pt->spp;
    wd.sampleBegin = rp->sample_begin;
    wd.totalWork = (unsigned long long)wd.nTilesShard * 256ull * (unsigned long long)wd.spp;
    int runCap = 256;  // (round 3, with the dense film flush: 64 -> 256 takes 5 % off k_generate, killeroo +0.9 %, the 10M-triangle scene
                       // +0.7 %, cornell-glass the same; 1024 measures as 256. Same-box A/B at the end of round 2: 16 -> 64 took 4 % off k_generate and 1 % off the camera-ray
                       // traversal, the other kernels unchanged -- 3596 -> 3620 Mray/s; 128 ... 1024 measure the same as 64.
                       // Earlier in the round longer runs cost the shading kernels 2-5 % and 16 was the optimum.)
    if (const char *e = getenv("MIPT_WORK_RUN")) runCap = std::max(1, atoi(e));
    wd.run = 1;
    while (2 * wd.run <= runCap && wd.spp % (2 * wd.run) == 0) wd.run *= 2;
    sub.iterations = 0;
    for (double &t : sub.t) t = 0;
    sub.result = DevCounters{};
    if (wd.totalWork == 0) return MI_OK;
    // Default pool: one slot per sample to render, up to the cap (earlier in round 2, at half: a 1/8 shard took 0.1196 s with the
    // quarter's 16M slots, 0.1169 s with 32M), at most 96M slots in total (below)
    // and at least 4M (8M per sub-renderer when several share the GPU): bigger pools mean fewer, better-filled
    // launches, but the last iterations of a render drain the pool at low occupancy, which a small job (one
    // shard of a multi-GPU frame) feels. Measured on the 1024-spp killeroo frame and its shards
    // (tools/shard_tune.py): a 1/8 shard on four sub-renderers takes 0.149 s with 16M slots and 0.143 s with
    // 32M; 64M slots (47 GB) make the traversal launches 3 % faster and k_shade / k_generate 4-6 % slower
    // (the path state outgrows the TLB reach), no gain for the full frame.
    uint32_t poolN = rp->path_pool;
    if (poolN == 0) {
        const unsigned long long quarter = wd.totalWork * (unsigned long long)subCount;   // (all of them: a 1/8 shard of the killeroo frame takes 0.1071 s on 32M slots, 0.1046 s on 64M)
        const unsigned long long floorN = subCount > 1 ? (8ull << 20) * (unsigned long long)subCount : (1ull << 22);
        // (`quarter`: half, since round 2.) The cap: 96M slots = 77 GB of path state of the 288 GB. With the kernels of the end
        // of round 2 bigger pools pay again (fewer, longer launches: the persistent traversal kernels lose less to their
        // tails, k_shade and k_generate no longer slow down): killeroo 1024 spp 3740 Mray/s at 32M, 3880 at 64M, 3912 at
        // 96M, 3931 at 128M; the 10M-triangle scene +0.4 %, cornell-glass -0.3 % at 64M.
        poolN = (uint32_t)std::min<unsigned long long>(3ull << 25, std::max<unsigned long long>(floorN, quarter));
    }
    poolN = std::max<uint32_t>(BLOCK, poolN / subCount / BLOCK * BLOCK);
    if (rp->path_pool == 0) {
        // the default is a preference: it takes at most 65 % of what the device has free (counting what this sub-renderer's
        // present pool would give back), so that what comes after the pool -- the film staging of the hand-over, RCCL's
        // buffers at the first collective, another renderer in the process -- still finds memory
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const size_t budget = (size_t)(0.65 * (double)(freeB + sub.poolBytes)) / (size_t)subCount;
            const size_t fit = budget / PoolSlotBytes(Q_COUNT + (s.nBands > 1 ? NQ : 0)) / BLOCK * BLOCK;
            if (fit < poolN) poolN = (uint32_t)std::max<size_t>(fit, (size_t)BLOCK);
        } else (void)hipGetLastError();
    }
    if (wd.totalWork < poolN) poolN = (uint32_t)((wd.totalWork + BLOCK - 1) / BLOCK * BLOCK);
    if (poolN < BLOCK) poolN = BLOCK;
    int rc = EnsurePool(sub, poolN, Q_COUNT + (s.nBands > 1 ? NQ : 0));
    // the default size is a preference, not a requirement: on a device that cannot hold it the render goes on with half,
    // a quarter, ... (an explicit mi_render_params.path_pool is taken at its word and fails)
    while (rc == MI_ERR_NOMEM && rp->path_pool == 0 && poolN > (1u << 22)) {
        poolN = poolN / 2 / BLOCK * BLOCK;
        rc = EnsurePool(sub, poolN, Q_COUNT + (s.nBands > 1 ? NQ : 0));
    }
    if (rc != MI_OK) return rc;
    HIPCHK(hipMemsetAsync(sub.pool.i + (size_t)I_FLAGS * poolN, 0, (size_t)poolN * sizeof(int), st));
    HIPCHK(hipMemsetAsync(sub.ctr, 0, sizeof(DevCounters), st));
    const dim3 grid((poolN + BLOCK - 1) / BLOCK), block(BLOCK);
    const dim3 chunkGrid((grid.x + SLOT_CHUNKS - 1) / SLOT_CHUNKS);   // kernels that take SLOT_CHUNKS x 256 slots per block
    const dim3 travGrid(std::min<unsigned>(grid.x, (unsigned)pt->numCUs * TRAV_BLOCKS_PER_CU));
    unsigned alive = 1;
    unsigned long long drawn = 0;
    // Per-kernel-class time: HIP events at the kernel boundaries of every iteration,
    // read back after the per-iteration sync that the alive counter needs anyway.
    auto harvest = [&](int set, bool full) {
        float ms = 0;
        hipEventElapsedTime(&ms, sub.evIter[set][0], sub.evIter[set][1]); sub.t[1] += ms * 1e-3;
        if (!full) return;
        hipEventElapsedTime(&ms, sub.evIter[set][1], sub.evIter[set][2]); sub.t[2] += ms * 1e-3;
        hipEventElapsedTime(&ms, sub.evIter[set][2], sub.evIter[set][3]); sub.t[3] += ms * 1e-3;
        hipEventElapsedTime(&ms, sub.evIter[set][3], sub.evIter[set][4]); sub.t[4] += ms * 1e-3;
        hipEventElapsedTime(&ms, sub.evIter[set][4], sub.evIter[set][5]); sub.t[5] += ms * 1e-3;
        // the closest-hit traversal launch alone: from an event recorded right before it (after the host's
        // per-iteration read of `alive`), so the host round trip is in [2] but not in [6]
        hipEventElapsedTime(&ms, sub.evIter[set][7], sub.evIter[set][6]); sub.t[6] += ms * 1e-3;
    };
    int set = 0;
    bool prevFull = false, havePrev = false;
    while (true) {
        hipEvent_t *ev = sub.evIter[set];
        HIPCHK(hipMemsetAsync(&sub.ctr->alive, 0, ITER_CLEAR_BYTES, st));
        HIPCHK(hipEventRecord(ev[0], st));
        hipLaunchKernelGGL(k_generate, chunkGrid, block, 0, st, s, sub.pool, pt->film, sub.ctr, wd);
        HIPCHK(hipEventRecord(ev[1], st));
        HIPCHK(hipMemcpyAsync(&alive, &sub.ctr->alive.v, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&drawn, &sub.ctr->nextWork, sizeof(drawn), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (havePrev) harvest(set ^ 1, prevFull);
        // done when no path is alive and every work item has been drawn (an iteration can draw nothing but items outside
        // the pixel bounds and leave the pool empty with work still to hand out)
        if (alive == 0 && drawn >= wd.totalWork) { harvest(set, false); break; }
        if (alive == 0) { harvest(set, false); havePrev = false; set ^= 1; if (++sub.iterations > 100000000ull) { g_err = "render loop did not terminate"; return MI_ERR_HIP; } continue; }
        HIPCHK(hipEventRecord(ev[7], st));
#ifdef MIPT_SORT_EXPERIMENT
        SortQueueExperiment(pt, sub, 0);
#endif
        LaunchTraversal(pt, sub, 0, travGrid);
        HIPCHK(hipEventRecord(ev[6], st));
        if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_extend<true>), chunkGrid, block, 0, st, s, sub.pool, sub.ctr);
        else hipLaunchKernelGGL((k_resolve_extend<false>), chunkGrid, block, 0, st, s, sub.pool, sub.ctr);
        if (pt->hasQuadrics) { if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_overflow<true>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 0);
            else hipLaunchKernelGGL((k_resolve_overflow<false>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 0); }
        HIPCHK(hipEventRecord(ev[2], st));
        LaunchShade(pt, sub, grid);
        HIPCHK(hipEventRecord(ev[3], st));
#ifdef MIPT_SORT_EXPERIMENT
        SortQueueExperiment(pt, sub, 1);
#endif
        LaunchTraversal(pt, sub, 1, travGrid);
        if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_shadow<true>), grid, block, 0, st, s, sub.pool, sub.ctr);
        else hipLaunchKernelGGL((k_resolve_shadow<false>), grid, block, 0, st, s, sub.pool, sub.ctr);
        if (pt->hasQuadrics) { if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_overflow<true>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 1);
            else hipLaunchKernelGGL((k_resolve_overflow<false>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 1); }
        HIPCHK(hipEventRecord(ev[4], st));
#ifdef MIPT_SORT_EXPERIMENT
        SortQueueExperiment(pt, sub, 2);
#endif
        LaunchTraversal(pt, sub, 2, travGrid);
        if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_mis<true>), grid, block, 0, st, s, sub.pool, sub.ctr);
        else hipLaunchKernelGGL((k_resolve_mis<false>), grid, block, 0, st, s, sub.pool, sub.ctr);
        if (pt->hasQuadrics || s.misAny) { if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_overflow<true>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 2);
            else hipLaunchKernelGGL((k_resolve_overflow<false>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 2); }
        HIPCHK(hipEventRecord(ev[5], st));
        HIPCHK(hipGetLastError());   // a launch of this iteration that was refused (bad configuration) stops the render here
        havePrev = true; prevFull = true;
        set ^= 1;
        if (++sub.iterations > 100000000ull) { g_err = "render loop did not terminate"; return MI_ERR_HIP; }
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(&sub.result, sub.ctr, sizeof(DevCounters), hipMemcpyDeviceToHost));
#ifdef MIPT_EXP_STAMPS
    {
        unsigned long long tot = 0;
        for (int k = 0; k < 19; ++k) tot += sub.result.phase[k];
        {
            unsigned long long g = 0;
            for (int k = 19; k < 24; ++k) g += sub.result.phase[k];
            fprintf(stderr, "k_generate stamps (scan + lists | L read + guards | film rows | refill | extend lists):");
            for (int k = 19; k < 24; ++k) fprintf(stderr, " %.1f%%", 100.0 * (double)sub.result.phase[k] / (double)std::max(1ull, g));
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "k_shade stamps: %llu waves, %.0f cycles per wave;", sub.result.phaseWaves, (double)tot / (double)std::max(1ull, sub.result.phaseWaves));
        for (int k = 0; k < 19; ++k) fprintf(stderr, " [%d] %.1f%%", k, 100.0 * (double)sub.result.phase[k] / (double)std::max(1ull, tot));
        fprintf(stderr, "\n");
    }
#endif
    return MI_OK;
}

int mi_pt_render(mi_pt *pt, const mi_render_params *rp, float *film_sum, float *weight_sum, mi_counters *counters) {
    if (!pt || !rp) { g_err = "null argument"; return MI_ERR_INVALID; }
    if (rp->shard_count < 1 || rp->shard_index < 0 || rp->shard_index >= rp->shard_count) { g_err = "bad shard"; return MI_ERR_INVALID; }
    if (pt->scene.samplerType >= MI_SAMPLER_ZEROTWO &&
        (rp->sample_begin < 0 || rp->sample_begin + (rp->spp_override > 0 ? rp->spp_override : pt->spp) > pt->spp)) {
        g_err = "a pixel sampler (02sequence / stratified) has tables for samples_per_pixel samples: the pass asks for sample numbers beyond them";
        return MI_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(pt->device));
    {   // what this pass lets the kernels leave out (DScene::index32 / storePixelSample)
        DScene &sc = pt->scene;
        const unsigned long long lastSample = (unsigned long long)rp->sample_begin + (unsigned long long)(rp->spp_override > 0 ? rp->spp_override : pt->spp);
        sc.index32 = (sc.samplerType == MI_SAMPLER_HALTON && (lastSample + 1ull) * (unsigned long long)std::max(1, sc.sampleStride) < (1ull << 32)) ? 1 : 0;
        sc.storePixelSample = (sc.samplerType >= MI_SAMPLER_RANDOM || sc.nBands > 1 || (pt->nTextures > 0 && sc.camera.lens_radius > 0)) ? 1 : 0;
    }
    hipStream_t st = (hipStream_t)rp->stream;
    if (!(rp->flags & MI_RENDER_ACCUMULATE)) HIPCHK(hipMemsetAsync(pt->film, 0, pt->nPix * 32 * sizeof(float), st));
    HIPCHK(hipStreamSynchronize(st));
    if (!(rp->flags & MI_RENDER_FILM_ON_DEVICE)) {   // the hand-over's staging before the pools size themselves
        if (film_sum && !pt->stageSum) HIPCHK(hipMalloc((void **)&pt->stageSum, pt->nPix * 31 * sizeof(float)));
        if (weight_sum && !pt->stageW) HIPCHK(hipMalloc((void **)&pt->stageW, pt->nPix * sizeof(float)));
    }
    const int nSub = (int)pt->subs.size();
    std::vector<int> rcs(nSub, MI_OK);
    std::vector<std::string> errs(nSub);
    const auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> threads;
        for (int k = 1; k < nSub; ++k)
            threads.emplace_back([&, k] { rcs[k] = RenderSub(pt, pt->subs[k], rp, k, nSub); errs[k] = g_err; });
        rcs[0] = RenderSub(pt, pt->subs[0], rp, 0, nSub);
        errs[0] = g_err;
        for (auto &t : threads) t.join();
    }
    for (int k = 0; k < nSub; ++k)
        if (rcs[k] != MI_OK) {
            // a failed sub-renderer may have left kernels queued: nothing of the caller's may be touched after we return
            (void)hipDeviceSynchronize();
            g_err = errs[k];
            return rcs[k];
        }
    HIPCHK(hipDeviceSynchronize());
    pt->lastSeconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long iterations = 0;
    DevStats c{};
    for (int i = 1; i < 7; ++i) pt->lastSeconds[i] = 0;
    for (const SubRenderer &sub : pt->subs) {
        for (int i = 1; i < 7; ++i) pt->lastSeconds[i] += sub.t[i];
        iterations += sub.iterations;
        for (const DevStats &r : sub.result.stats) {
            c.cameraRays += r.cameraRays; c.regularRays += r.regularRays; c.shadowRays += r.shadowRays;
            c.totalPaths += r.totalPaths; c.zeroRadiancePaths += r.zeroRadiancePaths; c.pathLengthSum += r.pathLengthSum;
            c.nodesVisited += r.nodesVisited; c.triTests += r.triTests; c.badSamples += r.badSamples;
            c.extendNodes += r.extendNodes; c.extendTris += r.extendTris; c.extendRays += r.extendRays;
        }
    }
    if (counters) {
        *counters = mi_counters{};
        counters->camera_rays = c.cameraRays; counters->regular_rays = c.regularRays; counters->shadow_rays = c.shadowRays;
        counters->total_paths = c.totalPaths; counters->zero_radiance_paths = c.zeroRadiancePaths;
        counters->path_length_sum = c.pathLengthSum; counters->bvh_nodes_visited = c.nodesVisited;
        counters->tri_tests = c.triTests; counters->bad_samples = c.badSamples;
        counters->iterations = iterations;
        counters->extend_rays = c.extendRays;
        counters->extend_nodes = c.extendNodes; counters->extend_tri_tests = c.extendTris;
        counters->launches[0] = counters->launches[1] = counters->launches[2] = iterations;
    }
    if (film_sum || weight_sum) {
        const bool onDev = (rp->flags & MI_RENDER_FILM_ON_DEVICE) != 0;
        float *dSum = nullptr, *dW = nullptr;
        if (onDev) { dSum = film_sum; dW = weight_sum; }
        else {   // staging for the host hand-over: allocated once per renderer (above), freed by mi_pt_destroy
            if (film_sum) dSum = pt->stageSum;
            if (weight_sum) dW = pt->stageW;
        }
        size_t total = pt->nPix * 32;
        hipLaunchKernelGGL(k_film_split, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pt->film, dSum, dW, pt->nPix);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        if (!onDev) {
            if (film_sum) HIPCHK(hipMemcpy(film_sum, dSum, pt->nPix * 31 * sizeof(float), hipMemcpyDeviceToHost));
            if (weight_sum) HIPCHK(hipMemcpy(weight_sum, dW, pt->nPix * sizeof(float), hipMemcpyDeviceToHost));
        }
    }
    return MI_OK;
}

int mi_pt_device_film(mi_pt *pt, void **dev_ptr, uint64_t *n_floats) {
    if (!pt || !dev_ptr || !n_floats) { g_err = "null argument"; return MI_ERR_INVALID; }
    *dev_ptr = pt->film;
    *n_floats = pt->nPix * 32;
    return MI_OK;
}

int mi_pt_pool_info(mi_pt *pt, uint64_t *slots, uint64_t *bytes) {
    if (!pt || !slots || !bytes) { g_err = "null argument"; return MI_ERR_INVALID; }
    *slots = 0; *bytes = 0;
    for (const SubRenderer &sub : pt->subs) { *slots += sub.pool.n; *bytes += sub.poolBytes; }
    return MI_OK;
}

int mi_pt_last_timings(mi_pt *pt, double *seconds, int n) {
    if (!pt || !seconds) { g_err = "null argument"; return MI_ERR_INVALID; }
    for (int i = 0; i < n && i < 8; ++i) seconds[i] = pt->lastSeconds[i];
    return MI_OK;
}

int mi_pt_texture_lookup(mi_pt *pt, int32_t tex, uint32_t n, const float *queries, float *rgb) {
    if (!pt || !queries || !rgb) { g_err = "null argument"; return MI_ERR_INVALID; }
    if (tex < 0 || (uint32_t)tex >= pt->nTextures) { g_err = "texture index out of range"; return MI_ERR_INVALID; }
    if (pt->textureTypes[tex] != MI_TEX_IMAGEMAP) { g_err = "mi_pt_texture_lookup: not an image texture"; return MI_ERR_INVALID; }
    if (n == 0) return MI_OK;
    HIPCHK(hipSetDevice(pt->device));
    DevBuf dq, dout;
    HIPCHK(dq.alloc((size_t)n * 6 * sizeof(float)));
    HIPCHK(dout.alloc((size_t)n * 3 * sizeof(float)));
    HIPCHK(hipMemcpy(dq.p, queries, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_texture_lookup, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, pt->scene, tex, dq.as<float>(), n, dout.as<float>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(rgb, dout.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_pt_light_distribution(mi_pt *pt, float *func, float *func_int, uint64_t capacity_voxels) {
    if (!pt) { g_err = "null argument"; return MI_ERR_INVALID; }
    const DScene &s = pt->scene;
    if (s.ldType != MI_LD_SPATIAL || s.nLights == 0) { g_err = "mi_pt_light_distribution: the scene has no spatial light distribution"; return MI_ERR_INVALID; }
    const size_t nVox = (size_t)s.nVoxels[0] * s.nVoxels[1] * s.nVoxels[2];
    if (capacity_voxels < nVox) { g_err = "mi_pt_light_distribution: buffer too small"; return MI_ERR_INVALID; }
    HIPCHK(hipSetDevice(pt->device));
    if (func) HIPCHK(hipMemcpy(func, s.ldFunc, nVox * s.nLights * sizeof(float), hipMemcpyDeviceToHost));
    if (func_int) HIPCHK(hipMemcpy(func_int, s.ldFuncInt, nVox * sizeof(float), hipMemcpyDeviceToHost));
    return MI_OK;
}

// Parity tool: one camera sample through the wavefront pipeline with a 256-slot pool, the path's state copied out at the
// kernel boundaries of every iteration (layout of a record: include/mi_pt.h, MI_PATH_RECORD_FLOATS).
int mi_pt_debug_path(mi_pt *pt, int32_t px, int32_t py, int64_t sample, int32_t max_records, float *records, int32_t *n_records) {
    if (!pt || !records || !n_records || max_records < 1) { g_err = "null argument"; return MI_ERR_INVALID; }
    *n_records = 0;
    HIPCHK(hipSetDevice(pt->device));
    const DScene saved = pt->scene;
    struct Restore { mi_pt *pt; DScene s; ~Restore() { pt->scene = s; } } restore{pt, saved};
    DScene &s = pt->scene;
    if (px < s.sampleBounds[0] || px >= s.sampleBounds[2] || py < s.sampleBounds[1] || py >= s.sampleBounds[3]) { g_err = "pixel outside the sample bounds"; return MI_ERR_INVALID; }
    s.pixelBounds[0] = px; s.pixelBounds[1] = py; s.pixelBounds[2] = px + 1; s.pixelBounds[3] = py + 1;
    s.index32 = 0; s.storePixelSample = 1;
    SubRenderer &sub = pt->subs[0];
    hipStream_t st = sub.stream;
    WorkDesc wd{};
    wd.nTilesX = (s.sampleBounds[2] - s.sampleBounds[0] + 15) / 16;
    wd.nTilesY = (s.sampleBounds[3] - s.sampleBounds[1] + 15) / 16;
    wd.shardIndex = ((py - s.sampleBounds[1]) / 16) * wd.nTilesX + (px - s.sampleBounds[0]) / 16;   // the pixel's tile alone
    wd.shardCount = wd.nTilesX * wd.nTilesY;
    wd.nTilesShard = 1;
    wd.spp = 1;
    wd.run = 1;
    wd.sampleBegin = sample;
    wd.totalWork = 256;
    const uint32_t poolN = BLOCK;
    int rc = EnsurePool(sub, poolN, Q_COUNT + (s.nBands > 1 ? NQ : 0));
    if (rc != MI_OK) return rc;
    HIPCHK(hipMemsetAsync(sub.pool.i + (size_t)I_FLAGS * poolN, 0, (size_t)poolN * sizeof(int), st));
    HIPCHK(hipMemsetAsync(sub.ctr, 0, sizeof(DevCounters), st));
    HIPCHK(hipMemsetAsync(pt->film, 0, pt->nPix * 32 * sizeof(float), st));
    const dim3 grid(1), block(BLOCK), travGrid(1);
    const Pool &pool = sub.pool;
    std::vector<int> flags(poolN);
    auto F4 = [&](int plane, uint32_t slot, float *dst) {   // Pool::R
        const float4 *src = plane < R_HIT ? pool.r + ((((size_t)(plane >> 1) * poolN + slot) << 1) + (plane & 1)) : pool.r + (size_t)plane * poolN + slot;
        return hipMemcpy(dst, src, 16, hipMemcpyDeviceToHost);
    };
    auto I1 = [&](int plane, uint32_t slot, int *dst) { return hipMemcpy(dst, pool.i + (size_t)plane * poolN + slot, 4, hipMemcpyDeviceToHost); };
    auto Spec = [&](int set, uint32_t slot, float *dst31) {
        float line[32];
        hipError_t e = hipMemcpy(line, pool.q + (((size_t)(set >> 3) * poolN + slot) << 3), 128, hipMemcpyDeviceToHost);
        memcpy(dst31, line, 31 * sizeof(float));
        return e;
    };
    int slot = -1;
    for (int it = 0; it < 4096; ++it) {
        HIPCHK(hipMemsetAsync(&sub.ctr->alive, 0, ITER_CLEAR_BYTES, st));
        hipLaunchKernelGGL(k_generate, grid, block, 0, st, s, sub.pool, pt->film, sub.ctr, wd);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(flags.data(), pool.i + (size_t)I_FLAGS * poolN, poolN * sizeof(int), hipMemcpyDeviceToHost));
        slot = -1;
        for (uint32_t k = 0; k < poolN; ++k) if (flags[k] & F_ALIVE) { slot = (int)k; break; }
        if (slot < 0) break;
        if (*n_records >= max_records) break;
        float *r = records + (size_t)(*n_records) * MI_PATH_RECORD_FLOATS;
        for (int k = 0; k < MI_PATH_RECORD_FLOATS; ++k) r[k] = 0.f;
        LaunchTraversal(pt, sub, 0, travGrid);
        if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_extend<true>), dim3(1), block, 0, st, s, sub.pool, sub.ctr);
        else hipLaunchKernelGGL((k_resolve_extend<false>), dim3(1), block, 0, st, s, sub.pool, sub.ctr);
        if (pt->hasQuadrics) { if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_overflow<true>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 0);
            else hipLaunchKernelGGL((k_resolve_overflow<false>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 0); }
        HIPCHK(hipStreamSynchronize(st));
        int bounces = 0, prim = -1, dim = 0;
        float ray0[4], ray1[4], hit[4];
        int word0 = 0;
        HIPCHK(I1(I_FLAGS, slot, &word0)); HIPCHK(I1(I_HITPRIM, slot, &prim));
        bounces = (word0 >> BOUNCE_SHIFT) & 0xff;
        if (s.samplerType >= MI_SAMPLER_ZEROTWO) HIPCHK(I1(I_DIM, slot, &dim)); else dim = (int)((unsigned)word0 >> DIM_SHIFT);
        HIPCHK(F4(R_RAY0, slot, ray0)); HIPCHK(F4(R_RAY1, slot, ray1)); HIPCHK(F4(R_HIT, slot, hit));
        r[0] = (float)bounces; r[1] = (float)prim; r[2] = (float)dim;
        r[4] = ray0[0]; r[5] = ray0[1]; r[6] = ray0[2]; r[7] = prim >= 0 ? hit[0] : ray0[3];
        r[8] = ray1[0]; r[9] = ray1[1]; r[10] = ray1[2]; r[11] = ray1[3];
        LaunchShade(pt, sub, grid);
        LaunchTraversal(pt, sub, 1, travGrid);
        if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_shadow<true>), grid, block, 0, st, s, sub.pool, sub.ctr);
        else hipLaunchKernelGGL((k_resolve_shadow<false>), grid, block, 0, st, s, sub.pool, sub.ctr);
        if (pt->hasQuadrics) { if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_overflow<true>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 1);
            else hipLaunchKernelGGL((k_resolve_overflow<false>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 1); }
        LaunchTraversal(pt, sub, 2, travGrid);
        if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_mis<true>), grid, block, 0, st, s, sub.pool, sub.ctr);
        else hipLaunchKernelGGL((k_resolve_mis<false>), grid, block, 0, st, s, sub.pool, sub.ctr);
        if (pt->hasQuadrics || s.misAny) { if (pt->hasInstances) hipLaunchKernelGGL((k_resolve_overflow<true>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 2);
            else hipLaunchKernelGGL((k_resolve_overflow<false>), dim3(OVERFLOW_GRID), block, 0, st, s, sub.pool, sub.ctr, 2); }
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        int fl = 0;
        HIPCHK(I1(I_FLAGS, slot, &fl));
        if (s.samplerType >= MI_SAMPLER_ZEROTWO) HIPCHK(I1(I_DIM, slot, &dim)); else dim = (int)((unsigned)fl >> DIM_SHIFT);
        HIPCHK(F4(R_RAY0, slot, ray0)); HIPCHK(F4(R_RAY1, slot, ray1));
        r[3] = (fl & F_ALIVE) ? 0.f : 1.f;
        r[12] = ray0[0]; r[13] = ray0[1]; r[14] = ray0[2]; r[15] = (float)dim;
        r[16] = ray1[0]; r[17] = ray1[1]; r[18] = ray1[2]; r[19] = ray1[3];
        if (fl & F_BETA_ONE) for (int k = 0; k < 31; ++k) r[20 + k] = 1.f;
        else HIPCHK(Spec(Q_BETA, slot, r + 20));
        if (!(fl & F_L_ZERO)) HIPCHK(Spec((fl & F_L_IN_B) ? Q_LB : Q_L, slot, r + 51));
        ++*n_records;
    }
    FreePool(sub.pool);   // the next render sizes its own
    sub.poolQuadPlanes = 0;
    HIPCHK(hipMemset(pt->film, 0, pt->nPix * 32 * sizeof(float)));
    return MI_OK;
}

int mi_pt_trace(mi_pt *pt, const float *rays, uint32_t n, int any_hit, float *hits) {
    if (!pt || !rays || !hits) { g_err = "null argument"; return MI_ERR_INVALID; }
    if (n == 0) return MI_OK;
    HIPCHK(hipSetDevice(pt->device));
    DevBuf dr, dh;
    HIPCHK(dr.alloc((size_t)n * 7 * sizeof(float)));
    HIPCHK(dh.alloc((size_t)n * 4 * sizeof(float)));
    HIPCHK(hipMemcpy(dr.p, rays, (size_t)n * 7 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_trace, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, pt->scene, dr.as<float>(), n, any_hit, dh.as<float>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(hits, dh.p, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_pt_math_probe(int device_ordinal, int op, uint32_t n, const float *x, const float *y, float *out) {
    if (!x || !y || !out || op < 0 || op > 5) { g_err = "mi_pt_math_probe: bad argument"; return MI_ERR_INVALID; }
    if (n == 0) return MI_OK;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || device_ordinal < 0 || device_ordinal >= nDev) { g_err = "no HIP device available (this path has no CPU fallback)"; return MI_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(device_ordinal));
    DevBuf dx, dy, dout;
    HIPCHK(dx.alloc((size_t)n * 2 * sizeof(float)));
    HIPCHK(dy.alloc((size_t)n * 2 * sizeof(float)));
    HIPCHK(dout.alloc((size_t)n * 3 * sizeof(float)));
    HIPCHK(hipMemcpy(dx.p, x, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy.p, y, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_math_probe, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, op, n, dx.as<float>(), dy.as<float>(), dout.as<float>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_pt_trace_wavefront(mi_pt *pt, const float *rays, uint32_t n, int mode, float *hits, float *extra) {
    if (!pt || !rays || !hits) { g_err = "null argument"; return MI_ERR_INVALID; }
    if (mode < 0 || mode > 3) { g_err = "mi_pt_trace_wavefront: mode must be 0 (path rays), 1 (shadow rays), 2 (MIS rays) or 3 (MIS rays as visibility queries)"; return MI_ERR_INVALID; }
    if (mode == 3 && !pt->scene.misAny) { g_err = "mi_pt_trace_wavefront: mode 3 needs a scene without instances and without an alpha mask on an emitter's mesh (others keep the closest-hit form of the MIS rays)"; return MI_ERR_INVALID; }
    if (n == 0) return MI_OK;
    if (n > (1u << 24)) { g_err = "mi_pt_trace_wavefront: at most 16M rays per call"; return MI_ERR_INVALID; }
    // the kernels fix what the render fixes: a shadow ray ends at 1 - ShadowEpsilon (Interaction::SpawnRayTo), a BSDF-sampled
    // ray never ends (SpawnRay)
    for (uint32_t i = 0; i < n && mode == 3; ++i)
        if (!(rays[(size_t)i * 7 + 6] > 0)) { g_err = "mi_pt_trace_wavefront: mode 3 rays carry the end of the emitter's span, tMax > 0"; return MI_ERR_INVALID; }
    for (uint32_t i = 0; i < n && (mode == 1 || mode == 2); ++i) {
        const float tMax = rays[(size_t)i * 7 + 6];
        if (mode == 1 ? tMax != 1 - kShadowEpsilon : !std::isinf(tMax) || tMax < 0) {
            g_err = mode == 1 ? "mi_pt_trace_wavefront: shadow rays (mode 1) carry tMax = 1 - 0.0001f" : "mi_pt_trace_wavefront: MIS rays (mode 2) carry tMax = +infinity";
            return MI_ERR_INVALID;
        }
    }
    HIPCHK(hipSetDevice(pt->device));
    SubRenderer &sub = pt->subs[0];
    hipStream_t st = sub.stream;
    const DScene &s = pt->scene;
    const uint32_t poolN = (n + SLOT_CHUNKS * BLOCK - 1) / (SLOT_CHUNKS * BLOCK) * (SLOT_CHUNKS * BLOCK);
    int rc = EnsurePool(sub, poolN, Q_COUNT + (s.nBands > 1 ? NQ : 0));
    if (rc != MI_OK) return rc;
    struct PoolGuard { SubRenderer &sub; ~PoolGuard() { FreePool(sub.pool); sub.poolQuadPlanes = 0; } } guard{sub};   // the next render sizes its own
    DevBuf dr, dh, dx;
    HIPCHK(dr.alloc((size_t)n * 7 * sizeof(float)));
    HIPCHK(dh.alloc((size_t)n * 4 * sizeof(float)));
    HIPCHK(dx.alloc((size_t)n * 4 * sizeof(float)));
    HIPCHK(hipMemcpy(dr.p, rays, (size_t)n * 7 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(sub.ctr, 0, sizeof(DevCounters), st));
    const dim3 grid(poolN / BLOCK), block(BLOCK);
    const dim3 chunkGrid(grid.x / SLOT_CHUNKS);
    const dim3 travGrid(std::min<unsigned>(grid.x, (unsigned)pt->numCUs * TRAV_BLOCKS_PER_CU));
    hipLaunchKernelGGL(k_trace_load, grid, block, 0, st, sub.pool, sub.ctr, dr.as<float>(), n, mode);
    LaunchTraversal(pt, sub, mode, travGrid, true);   // (mode 2: the closest-hit kernel, whose records this call returns)
    hipLaunchKernelGGL(k_trace_raw, grid, block, 0, st, sub.pool, n, mode, dx.as<float>());
    const bool inst = pt->hasInstances;
#define MIPT_BY_INST(K, G, ...) do { if (inst) hipLaunchKernelGGL((K<true>), G, block, 0, st, __VA_ARGS__); else hipLaunchKernelGGL((K<false>), G, block, 0, st, __VA_ARGS__); } while (0)
    if (mode == 0) MIPT_BY_INST(k_resolve_extend, chunkGrid, s, sub.pool, sub.ctr);
    else if (mode == 1) MIPT_BY_INST(k_resolve_shadow, grid, s, sub.pool, sub.ctr);
    if (mode < 2 && pt->hasQuadrics) MIPT_BY_INST(k_resolve_overflow, dim3(OVERFLOW_GRID), s, sub.pool, sub.ctr, mode);
    MIPT_BY_INST(k_trace_read, grid, s, sub.pool, n, mode, dh.as<float>(), dx.as<float>());
#undef MIPT_BY_INST
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(hits, dh.p, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    if (extra) HIPCHK(hipMemcpy(extra, dx.p, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return MI_OK;
}

void mi_pt_destroy(mi_pt *pt) {
    if (!pt) return;
    hipSetDevice(pt->device);
    hipDeviceSynchronize();
    for (void *p : pt->allocs) hipFree(p);
    if (pt->film) hipFree(pt->film);
    if (pt->stageSum) hipFree(pt->stageSum);
    if (pt->stageW) hipFree(pt->stageW);
    for (SubRenderer &sub : pt->subs) {
        FreePool(sub.pool);
        if (sub.ctr) hipFree(sub.ctr);
        for (int a = 0; a < 2; ++a) for (int b = 0; b < N_EV; ++b) if (sub.evIter[a][b]) hipEventDestroy(sub.evIter[a][b]);
        if (sub.stream) hipStreamDestroy(sub.stream);
    }
    delete pt;
}

}  // extern "C"
#endif   // MIPT_HAS_MAIN
