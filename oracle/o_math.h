// oracle/o_math.h -- TEST INFRASTRUCTURE (CPU oracle). Not part of the product:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.
//
// Scalar/vector helpers restating the reference's float semantics exactly
// (src/core/pbrt.h:244-294,307-320,420; src/core/geometry.h:245-249,954-1036,
// 1222-1239,1449-1481). Compiled with -ffp-contract=off, no fast-math.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

typedef float Float;

// ---- libm. The reference calls std::sin(float) & co.; how their last bit is rounded is the C library's business
// (glibc 2.35 here: within an ulp, usually but not always the nearest float). Two evaluations are kept:
//   mode 0 (default)  the host's float functions as the reference binary calls them -- the mode every pin against the
//                     reference's own numbers runs in (BASELINE.md counters are reproduced exactly in it);
//   mode 1            the correctly rounded float result (evaluated in double, rounded once), which is what the
//                     device computes (d_math.h) -- device-vs-oracle parity tests run in it, so that a difference
//                     between the two is a defect and not a last-bit libm artefact amplified by the path.
// The two modes differ in O(1e-4) of the samples of a frame (tests/test_oracle_pins.py measures it).
inline int g_libmMode = 0;
inline Float SinF(Float x) { return g_libmMode ? (Float)std::sin((double)x) : std::sin(x); }
inline Float CosF(Float x) { return g_libmMode ? (Float)std::cos((double)x) : std::cos(x); }
inline Float AcosF(Float x) { return g_libmMode ? (Float)std::acos((double)x) : std::acos(x); }
inline Float Atan2F(Float y, Float x) { return g_libmMode ? (Float)std::atan2((double)y, (double)x) : std::atan2(y, x); }
inline Float LogF(Float x) { return g_libmMode ? (Float)std::log((double)x) : std::log(x); }
inline Float PowF(Float x, Float y) { return g_libmMode ? (Float)std::pow((double)x, (double)y) : std::pow(x, y); }

static constexpr Float Infinity = std::numeric_limits<Float>::infinity();
static constexpr Float MachineEpsilon = std::numeric_limits<Float>::epsilon() * 0.5;
static constexpr Float ShadowEpsilon = 0.0001f;
static constexpr Float Pi = 3.14159265358979323846;
static constexpr Float InvPi = 0.31830988618379067154;
static constexpr Float Inv2Pi = 0.15915494309189533577;  // pbrt.h:210
static constexpr Float Inv4Pi = 0.07957747154594766788;
static constexpr Float PiOver2 = 1.57079632679489661923;
static constexpr Float PiOver4 = 0.78539816339744830961;
static const Float OneMinusEpsilon = 0x1.fffffep-1;  // src/core/rng.h:51-57

inline Float gamma(int n) { return (n * MachineEpsilon) / (1 - n * MachineEpsilon); }
inline Float Lerp(Float t, Float v1, Float v2) { return (1 - t) * v1 + t * v2; }
template <typename T, typename U, typename V>
inline T Clamp(T val, U low, V high) {
    if (val < low) return low;
    else if (val > high) return high;
    else return val;
}
inline uint32_t FloatToBits(float f) { uint32_t ui; memcpy(&ui, &f, sizeof(float)); return ui; }
inline float BitsToFloat(uint32_t ui) { float f; memcpy(&f, &ui, sizeof(uint32_t)); return f; }
inline float NextFloatUp(float v) {  // pbrt.h:244-256
    if (std::isinf(v) && v > 0.) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = FloatToBits(v);
    if (v >= 0) ++ui; else --ui;
    return BitsToFloat(ui);
}
inline float NextFloatDown(float v) {  // pbrt.h:258-268
    if (std::isinf(v) && v < 0.) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = FloatToBits(v);
    if (v > 0) --ui; else ++ui;
    return BitsToFloat(ui);
}

struct V3 {
    Float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(Float x, Float y, Float z) : x(x), y(y), z(z) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    V3 operator+(const V3 &v) const { return V3(x + v.x, y + v.y, z + v.z); }
    V3 operator-(const V3 &v) const { return V3(x - v.x, y - v.y, z - v.z); }
    V3 operator*(Float s) const { return V3(x * s, y * s, z * s); }
    V3 operator-() const { return V3(-x, -y, -z); }
    V3 &operator+=(const V3 &v) { x += v.x; y += v.y; z += v.z; return *this; }
    V3 &operator*=(Float s) { x *= s; y *= s; z *= s; return *this; }
    V3 operator/(Float f) const { Float inv = (Float)1 / f; return V3(x * inv, y * inv, z * inv); }
    Float LengthSquared() const { return x * x + y * y + z * z; }
    Float Length() const { return std::sqrt(LengthSquared()); }
};
inline V3 operator*(Float s, const V3 &v) { return v * s; }
inline Float Dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float AbsDot(const V3 &a, const V3 &b) { return std::abs(Dot(a, b)); }
inline V3 Cross(const V3 &v1, const V3 &v2) {  // geometry.h:966-972: computed in double
    double v1x = v1.x, v1y = v1.y, v1z = v1.z;
    double v2x = v2.x, v2y = v2.y, v2z = v2.z;
    return V3((Float)((v1y * v2z) - (v1z * v2y)), (Float)((v1z * v2x) - (v1x * v2z)),
              (Float)((v1x * v2y) - (v1y * v2x)));
}
inline V3 Normalize(const V3 &v) { return v / v.Length(); }
inline V3 Abs(const V3 &v) { return V3(std::abs(v.x), std::abs(v.y), std::abs(v.z)); }
inline Float DistanceSquared(const V3 &a, const V3 &b) { return (a - b).LengthSquared(); }
inline Float Distance(const V3 &a, const V3 &b) { return (a - b).Length(); }
inline int MaxDimension(const V3 &v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
inline Float MaxComponent(const V3 &v) { return std::max(v.x, std::max(v.y, v.z)); }
inline V3 Permute(const V3 &v, int x, int y, int z) { return V3(v[x], v[y], v[z]); }
inline V3 Faceforward(const V3 &n, const V3 &v) { return (Dot(n, v) < 0.f) ? -n : n; }
inline void CoordinateSystem(const V3 &v1, V3 *v2, V3 *v3) {  // geometry.h:1029-1036
    if (std::abs(v1.x) > std::abs(v1.y))
        *v2 = V3(-v1.z, 0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else
        *v2 = V3(0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = Cross(v1, *v2);
}
inline V3 SphericalDirection(Float sinTheta, Float cosTheta, Float phi) {
    return V3(sinTheta * CosF(phi), sinTheta * SinF(phi), cosTheta);
}
inline V3 SphericalDirection(Float sinTheta, Float cosTheta, Float phi, const V3 &x, const V3 &y, const V3 &z) {
    return sinTheta * CosF(phi) * x + sinTheta * SinF(phi) * y + cosTheta * z;
}
inline V3 OffsetRayOrigin(const V3 &p, const V3 &pError, const V3 &n, const V3 &w) {  // geometry.h:1449-1469
    Float d = Dot(Abs(n), pError);
    V3 offset = d * n;
    if (Dot(w, n) < 0) offset = -offset;
    V3 po = p + offset;
    for (int i = 0; i < 3; ++i) {
        if (offset[i] > 0) po[i] = NextFloatUp(po[i]);
        else if (offset[i] < 0) po[i] = NextFloatDown(po[i]);
    }
    return po;
}

struct Ray {
    V3 o, d;
    mutable Float tMax;
    Ray() : tMax(Infinity) {}
    Ray(const V3 &o, const V3 &d, Float tMax = Infinity) : o(o), d(d), tMax(tMax) {}
    V3 operator()(Float t) const { return o + d * t; }
};

// 31-bin spectrum (src/core/spectrum.h:106-293)
static constexpr int NS = 31;
struct Spec {
    Float c[NS];
    Spec(Float v = 0.f) { for (int i = 0; i < NS; ++i) c[i] = v; }
    static Spec From(const float *v) { Spec s; for (int i = 0; i < NS; ++i) s.c[i] = v[i]; return s; }
    Spec &operator+=(const Spec &s) { for (int i = 0; i < NS; ++i) c[i] += s.c[i]; return *this; }
    Spec operator+(const Spec &s) const { Spec r = *this; for (int i = 0; i < NS; ++i) r.c[i] += s.c[i]; return r; }
    Spec operator-(const Spec &s) const { Spec r = *this; for (int i = 0; i < NS; ++i) r.c[i] -= s.c[i]; return r; }
    Spec operator*(const Spec &s) const { Spec r = *this; for (int i = 0; i < NS; ++i) r.c[i] *= s.c[i]; return r; }
    Spec &operator*=(const Spec &s) { for (int i = 0; i < NS; ++i) c[i] *= s.c[i]; return *this; }
    Spec operator*(Float a) const { Spec r = *this; for (int i = 0; i < NS; ++i) r.c[i] *= a; return r; }
    Spec &operator*=(Float a) { for (int i = 0; i < NS; ++i) c[i] *= a; return *this; }
    Spec operator/(Float a) const { Spec r = *this; for (int i = 0; i < NS; ++i) r.c[i] /= a; return r; }
    Spec operator/(const Spec &s) const { Spec r = *this; for (int i = 0; i < NS; ++i) r.c[i] /= s.c[i]; return r; }
    Spec &operator/=(Float a) { for (int i = 0; i < NS; ++i) c[i] /= a; return *this; }
    bool IsBlack() const { for (int i = 0; i < NS; ++i) if (c[i] != 0.) return false; return true; }
    Float MaxComponentValue() const { Float m = c[0]; for (int i = 1; i < NS; ++i) m = std::max(m, c[i]); return m; }
    bool HasNaNs() const { for (int i = 0; i < NS; ++i) if (std::isnan(c[i])) return true; return false; }
};
inline Spec operator*(Float a, const Spec &s) { return s * a; }
inline Spec Sqrt(const Spec &s) { Spec r; for (int i = 0; i < NS; ++i) r.c[i] = std::sqrt(s.c[i]); return r; }
inline Spec Lerp(Float t, const Spec &s1, const Spec &s2) { return (1 - t) * s1 + t * s2; }

}  // namespace orc
