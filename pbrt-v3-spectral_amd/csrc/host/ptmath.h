// ptmath.h -- host-side POD math for the scene front end (C++17, float32).
//
// Semantics follow the reference's geometry/transform layer so that the flattened
// scene (world-space vertices, BVH bounds, camera matrices) comes out with the same
// bits: src/core/geometry.h (Vector3/Point3/Normal3/Bounds3), src/core/transform.h
// and transform.cpp (Matrix4x4, Transform, LookAt, Perspective ...), src/core/pbrt.h
// (gamma, NextFloatUp/Down, Radians). Types here are plain structs, not a port of
// the template hierarchy. Build with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>

namespace mipt {

static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 0.31830988618379067154f;
static constexpr float kInfinity = std::numeric_limits<float>::infinity();
static constexpr float kMachineEpsilon = std::numeric_limits<float>::epsilon() * 0.5f;

inline float gammaf(int n) {  // src/core/pbrt.h:292-294
    return (n * kMachineEpsilon) / (1 - n * kMachineEpsilon);
}
inline float Radians(float deg) { return (kPi / 180) * deg; }  // pbrt.h:327
inline float Lerp(float t, float v1, float v2) { return (1 - t) * v1 + t * v2; }
template <typename T, typename U, typename V>
inline T Clamp(T val, U low, V high) {
    if (val < low) return low;
    else if (val > high) return high;
    else return val;
}

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x, float y, float z) : x(x), y(y), z(z) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    Vec3 operator+(const Vec3 &v) const { return Vec3(x + v.x, y + v.y, z + v.z); }
    Vec3 operator-(const Vec3 &v) const { return Vec3(x - v.x, y - v.y, z - v.z); }
    Vec3 operator*(float s) const { return Vec3(x * s, y * s, z * s); }
    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    Vec3 &operator+=(const Vec3 &v) { x += v.x; y += v.y; z += v.z; return *this; }
    // geometry.h:245-249 -- division multiplies by the reciprocal
    Vec3 operator/(float f) const { float inv = (float)1 / f; return Vec3(x * inv, y * inv, z * inv); }
    bool operator==(const Vec3 &v) const { return x == v.x && y == v.y && z == v.z; }
    float LengthSquared() const { return x * x + y * y + z * z; }
    float Length() const { return std::sqrt(LengthSquared()); }
};
inline Vec3 operator*(float s, const Vec3 &v) { return v * s; }
inline float Dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 Cross(const Vec3 &v1, const Vec3 &v2) {  // geometry.h:966-972 (double)
    double v1x = v1.x, v1y = v1.y, v1z = v1.z;
    double v2x = v2.x, v2y = v2.y, v2z = v2.z;
    return Vec3((float)((v1y * v2z) - (v1z * v2y)), (float)((v1z * v2x) - (v1x * v2z)),
                (float)((v1x * v2y) - (v1y * v2x)));
}
inline Vec3 Normalize(const Vec3 &v) { return v / v.Length(); }
inline Vec3 Min(const Vec3 &a, const Vec3 &b) {
    return Vec3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z));
}
inline Vec3 Max(const Vec3 &a, const Vec3 &b) {
    return Vec3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z));
}

struct Bounds3 {  // geometry.h Bounds3f
    Vec3 pMin{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
              std::numeric_limits<float>::max()};
    Vec3 pMax{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
              std::numeric_limits<float>::lowest()};
    Bounds3() = default;
    explicit Bounds3(const Vec3 &p) : pMin(p), pMax(p) {}
    Bounds3(const Vec3 &a, const Vec3 &b) : pMin(Min(a, b)), pMax(Max(a, b)) {}
    Vec3 Diagonal() const { return pMax - pMin; }
    float SurfaceArea() const {
        Vec3 d = Diagonal();
        return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    int MaximumExtent() const {
        Vec3 d = Diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        else return 2;
    }
    Vec3 Offset(const Vec3 &p) const {
        Vec3 o = p - pMin;
        if (pMax.x > pMin.x) o.x /= pMax.x - pMin.x;
        if (pMax.y > pMin.y) o.y /= pMax.y - pMin.y;
        if (pMax.z > pMin.z) o.z /= pMax.z - pMin.z;
        return o;
    }
    Vec3 LerpP(const Vec3 &t) const {
        return Vec3(Lerp(t.x, pMin.x, pMax.x), Lerp(t.y, pMin.y, pMax.y), Lerp(t.z, pMin.z, pMax.z));
    }
    void BoundingSphere(Vec3 *c, float *rad) const {
        *c = (pMin + pMax) / 2;
        bool inside = c->x >= pMin.x && c->x <= pMax.x && c->y >= pMin.y && c->y <= pMax.y &&
                      c->z >= pMin.z && c->z <= pMax.z;
        *rad = inside ? (*c - pMax).Length() : 0;
    }
};
inline Bounds3 Union(const Bounds3 &b, const Vec3 &p) {
    Bounds3 r;
    r.pMin = Min(b.pMin, p);
    r.pMax = Max(b.pMax, p);
    return r;
}
inline Bounds3 Union(const Bounds3 &a, const Bounds3 &b) {
    Bounds3 r;
    r.pMin = Min(a.pMin, b.pMin);
    r.pMax = Max(a.pMax, b.pMax);
    return r;
}

struct Matrix4x4 {
    float m[4][4];
    Matrix4x4() {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) m[i][j] = (i == j) ? 1.f : 0.f;
    }
    Matrix4x4(float t00, float t01, float t02, float t03, float t10, float t11, float t12, float t13,
              float t20, float t21, float t22, float t23, float t30, float t31, float t32, float t33) {
        m[0][0] = t00; m[0][1] = t01; m[0][2] = t02; m[0][3] = t03;
        m[1][0] = t10; m[1][1] = t11; m[1][2] = t12; m[1][3] = t13;
        m[2][0] = t20; m[2][1] = t21; m[2][2] = t22; m[2][3] = t23;
        m[3][0] = t30; m[3][1] = t31; m[3][2] = t32; m[3][3] = t33;
    }
    bool operator==(const Matrix4x4 &o) const { return std::memcmp(m, o.m, sizeof(m)) == 0; }
    static Matrix4x4 Mul(const Matrix4x4 &m1, const Matrix4x4 &m2) {  // transform.h:86-93
        Matrix4x4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                r.m[i][j] = m1.m[i][0] * m2.m[0][j] + m1.m[i][1] * m2.m[1][j] +
                            m1.m[i][2] * m2.m[2][j] + m1.m[i][3] * m2.m[3][j];
        return r;
    }
};
Matrix4x4 Transpose(const Matrix4x4 &m);
Matrix4x4 Inverse(const Matrix4x4 &m, bool *singular = nullptr);  // transform.cpp:82-136

struct Transform {
    Matrix4x4 m, mInv;
    Transform() = default;
    explicit Transform(const Matrix4x4 &mm) : m(mm), mInv(Inverse(mm)) {}
    Transform(const Matrix4x4 &mm, const Matrix4x4 &mi) : m(mm), mInv(mi) {}
    Transform operator*(const Transform &t2) const {  // transform.cpp:244-246
        return Transform(Matrix4x4::Mul(m, t2.m), Matrix4x4::Mul(t2.mInv, mInv));
    }
    bool operator==(const Transform &t) const { return m == t.m && mInv == t.mInv; }
    bool operator!=(const Transform &t) const { return !(*this == t); }
    Vec3 Point(const Vec3 &p) const {  // transform.h:222-233
        float x = p.x, y = p.y, z = p.z;
        float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
        float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
        float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
        float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
        if (wp == 1) return Vec3(xp, yp, zp);
        float inv = (float)1 / wp;  // Point3::operator/ geometry.h:500-504
        return Vec3(inv * xp, inv * yp, inv * zp);
    }
    Vec3 Vector(const Vec3 &v) const {  // transform.h:236-241
        float x = v.x, y = v.y, z = v.z;
        return Vec3(m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z,
                    m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
                    m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z);
    }
    Vec3 Normal(const Vec3 &n) const {  // transform.h:244-249 (inverse transpose)
        float x = n.x, y = n.y, z = n.z;
        return Vec3(mInv.m[0][0] * x + mInv.m[1][0] * y + mInv.m[2][0] * z,
                    mInv.m[0][1] * x + mInv.m[1][1] * y + mInv.m[2][1] * z,
                    mInv.m[0][2] * x + mInv.m[1][2] * y + mInv.m[2][2] * z);
    }
    Bounds3 Bounds(const Bounds3 &b) const;  // transform.cpp:231-242
    bool SwapsHandedness() const {           // transform.cpp:248-253
        float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) -
                    m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
                    m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
        return det < 0;
    }
    bool HasScale() const {  // transform.h:155-162
        float la2 = Vector(Vec3(1, 0, 0)).LengthSquared();
        float lb2 = Vector(Vec3(0, 1, 0)).LengthSquared();
        float lc2 = Vector(Vec3(0, 0, 1)).LengthSquared();
#define MIPT_NOT_ONE(x) ((x) < .999f || (x) > 1.001f)
        return (MIPT_NOT_ONE(la2) || MIPT_NOT_ONE(lb2) || MIPT_NOT_ONE(lc2));
#undef MIPT_NOT_ONE
    }
};
inline void CoordinateSystem(const Vec3 &v1, Vec3 *v2, Vec3 *v3) {  // geometry.h:1029-1036
    if (std::abs(v1.x) > std::abs(v1.y)) *v2 = Vec3(-v1.z, 0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else *v2 = Vec3(0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = Cross(v1, *v2);
}
inline Transform Inverse(const Transform &t) { return Transform(t.mInv, t.m); }
Transform Translate(const Vec3 &delta);
Transform Scale(float x, float y, float z);
Transform Rotate(float theta, const Vec3 &axis);
Transform LookAt(const Vec3 &pos, const Vec3 &look, const Vec3 &up, bool *degenerate);
Transform Perspective(float fov, float n, float f);

}  // namespace mipt
