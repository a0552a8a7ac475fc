// d_math.h -- device-side scalar/vector helpers for the gfx950 path-tracing kernels.
// Float semantics follow the reference so device decisions match the CPU reference:
// no FMA contraction (-ffp-contract=off), IEEE divide/sqrt (hipcc default), Cross in
// double (src/core/geometry.h:966-972), vector division as multiply-by-reciprocal
// (geometry.h:245-249), NaN-transparent min/max written as comparisons (std::min /
// std::max semantics), error-bounded ray offsets (geometry.h:1449-1469).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__

namespace dpt {

static constexpr float kInfinity = __builtin_huge_valf();
static constexpr float kMachineEpsilon = 5.9604644775390625e-08f;  // epsilon * 0.5
static constexpr float kShadowEpsilon = 0.0001f;
static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 0.31830988618379067154f;
static constexpr float kPiOver2 = 1.57079632679489661923f;
static constexpr float kPiOver4 = 0.78539816339744830961f;
static constexpr float kOneMinusEpsilon = 0x1.fffffep-1f;

// gamma(n) = (n*eps)/(1-n*eps), src/core/pbrt.h:292-294, folded per n at compile time
DEV constexpr float gammaf(int n) { return ((float)n * kMachineEpsilon) / (1 - (float)n * kMachineEpsilon); }

DEV float minf(float a, float b) { return (b < a) ? b : a; }  // std::min
DEV float maxf(float a, float b) { return (a < b) ? b : a; }  // std::max
DEV float clampf(float v, float lo, float hi) { return (v < lo) ? lo : ((v > hi) ? hi : v); }
DEV float lerpf(float t, float v1, float v2) { return (1 - t) * v1 + t * v2; }
DEV float absf(float a) { return __builtin_fabsf(a); }
DEV bool isinff(float a) { return __builtin_isinf(a); }
DEV bool isnanf_(float a) { return a != a; }

// libm calls. The CPU reference uses glibc's float functions, which are correctly
// rounded in (nearly) all cases; ocml's float versions are only ~1-2 ulp. Evaluating
// in double and rounding once gives the correctly rounded float result (up to
// double-rounding ties), so a path takes the same discrete decisions as on the CPU.
// They are deliberately not inlined: each expands to several hundred instructions, and the shading kernels
// are bound by instruction fetch (their code does not fit the instruction cache), not by call overhead.
#define DEV_CALL __device__ __noinline__
#ifdef MIPT_EXP_FLOATLIBM   // (timing experiment: ocml's float routines, inlined)
DEV float sinF(float x) { return ::sinf(x); }
DEV float cosF(float x) { return ::cosf(x); }
DEV float acosF(float x) { return ::acosf(x); }
DEV float atan2F(float y, float x) { return ::atan2f(y, x); }
DEV float logF(float x) { return ::logf(x); }
DEV float powF(float x, float y) { return ::powf(x, y); }
#else
static DEV_CALL float sinF(float x) { return (float)sin((double)x); }
static DEV_CALL float cosF(float x) { return (float)cos((double)x); }
static DEV_CALL float acosF(float x) { return (float)acos((double)x); }
static DEV_CALL float atan2F(float y, float x) { return (float)atan2((double)y, (double)x); }
static DEV_CALL float logF(float x) { return (float)log((double)x); }
static DEV_CALL float powF(float x, float y) { return (float)pow((double)x, (double)y); }
#endif

// x / d for many x and one d: one IEEE reciprocal, then per quotient a multiply and two
// fused corrections (q = x*r; rem = fma(-q, d, x); q += rem*r). With r the correctly
// rounded reciprocal this is the correctly rounded quotient except when x/d lies within
// ~2^-23 ulp of a rounding boundary (about 2^-22 of operands, then off by one ulp);
// operands whose reciprocal or quotient leave the normal range use the plain division.
struct Divisor {
    float d, r;
    bool fast;
};
DEV Divisor MakeDivisor(float d) {
    Divisor v;
    v.d = d;
    v.r = 1.f / d;
    const float a = absf(d);
    v.fast = (a > 1e-18f) && (a < 1e18f);
    return v;
}
DEV float DivBy(float x, const Divisor &v) {
    if (v.fast) {
        const float q = x * v.r;
        const float rem = __builtin_fmaf(-q, v.d, x);
        const float q1 = __builtin_fmaf(rem, v.r, q);
        const float aq = absf(q1);
        if (aq > 1e-30f && aq < 1e30f) return q1;
    }
    return x / v.d;
}

DEV float NextFloatUp(float v) {  // pbrt.h:244-256
    if (isinff(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = __float_as_uint(v);
    if (v >= 0) ++ui; else --ui;
    return __uint_as_float(ui);
}
DEV float NextFloatDown(float v) {  // pbrt.h:258-268
    if (isinff(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = __float_as_uint(v);
    if (v > 0) --ui; else ++ui;
    return __uint_as_float(ui);
}

struct V3 {
    float x, y, z;
    DEV V3() : x(0), y(0), z(0) {}
    DEV V3(float x, float y, float z) : x(x), y(y), z(z) {}
    DEV float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    DEV V3 operator+(const V3 &v) const { return V3(x + v.x, y + v.y, z + v.z); }
    DEV V3 operator-(const V3 &v) const { return V3(x - v.x, y - v.y, z - v.z); }
    DEV V3 operator*(float s) const { return V3(x * s, y * s, z * s); }
    DEV V3 operator-() const { return V3(-x, -y, -z); }
    DEV V3 &operator+=(const V3 &v) { x += v.x; y += v.y; z += v.z; return *this; }
    DEV V3 &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    DEV V3 operator/(float f) const { float inv = 1.f / f; return V3(x * inv, y * inv, z * inv); }
    DEV float LengthSquared() const { return x * x + y * y + z * z; }
    DEV float Length() const { return __builtin_sqrtf(LengthSquared()); }
};
DEV V3 operator*(float s, const V3 &v) { return v * s; }
DEV float Dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV float AbsDot(const V3 &a, const V3 &b) { return absf(Dot(a, b)); }
DEV V3 Cross(const V3 &v1, const V3 &v2) {
    double v1x = v1.x, v1y = v1.y, v1z = v1.z;
    double v2x = v2.x, v2y = v2.y, v2z = v2.z;
    return V3((float)((v1y * v2z) - (v1z * v2y)), (float)((v1z * v2x) - (v1x * v2z)),
              (float)((v1x * v2y) - (v1y * v2x)));
}
DEV V3 Normalize(const V3 &v) { return v / v.Length(); }
DEV V3 Abs(const V3 &v) { return V3(absf(v.x), absf(v.y), absf(v.z)); }
DEV float DistanceSquared(const V3 &a, const V3 &b) { return (a - b).LengthSquared(); }
DEV float Distance(const V3 &a, const V3 &b) { return (a - b).Length(); }
DEV int MaxDimension(const V3 &v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
DEV float MaxComponent(const V3 &v) { return maxf(v.x, maxf(v.y, v.z)); }
DEV V3 Faceforward(const V3 &n, const V3 &v) { return (Dot(n, v) < 0.f) ? -n : n; }
DEV void CoordinateSystem(const V3 &v1, V3 *v2, V3 *v3) {  // geometry.h:1029-1036
    if (absf(v1.x) > absf(v1.y))
        *v2 = V3(-v1.z, 0, v1.x) / __builtin_sqrtf(v1.x * v1.x + v1.z * v1.z);
    else
        *v2 = V3(0, v1.z, -v1.y) / __builtin_sqrtf(v1.y * v1.y + v1.z * v1.z);
    *v3 = Cross(v1, *v2);
}
DEV V3 SphericalDirection(float sinTheta, float cosTheta, float phi) {
    return V3(sinTheta * cosF(phi), sinTheta * sinF(phi), cosTheta);
}
DEV V3 SphericalDirection(float sinTheta, float cosTheta, float phi, const V3 &x, const V3 &y, const V3 &z) {
    return sinTheta * cosF(phi) * x + sinTheta * sinF(phi) * y + cosTheta * z;
}
DEV V3 OffsetRayOrigin(const V3 &p, const V3 &pError, const V3 &n, const V3 &w) {
    float d = Dot(Abs(n), pError);
    V3 offset = d * n;
    if (Dot(w, n) < 0) offset = -offset;
    V3 po = p + offset;
    if (offset.x > 0) po.x = NextFloatUp(po.x); else if (offset.x < 0) po.x = NextFloatDown(po.x);
    if (offset.y > 0) po.y = NextFloatUp(po.y); else if (offset.y < 0) po.y = NextFloatDown(po.y);
    if (offset.z > 0) po.z = NextFloatUp(po.z); else if (offset.z < 0) po.z = NextFloatDown(po.z);
    return po;
}

struct Ray {
    V3 o, d;
    float tMax;
    DEV Ray() : tMax(kInfinity) {}
    DEV Ray(const V3 &o, const V3 &d, float tMax = kInfinity) : o(o), d(d), tMax(tMax) {}
    DEV V3 at(float t) const { return o + d * t; }
};

}  // namespace dpt
