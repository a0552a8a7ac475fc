// ptmath.cpp -- matrix inverse and the transform constructors the .pbrt
// directives need. Arithmetic order follows src/core/transform.cpp so CTMs (and
// therefore world-space vertices) match the reference.
#include "ptmath.h"

namespace mipt {

Matrix4x4 Transpose(const Matrix4x4 &m) {
    return Matrix4x4(m.m[0][0], m.m[1][0], m.m[2][0], m.m[3][0], m.m[0][1], m.m[1][1], m.m[2][1],
                     m.m[3][1], m.m[0][2], m.m[1][2], m.m[2][2], m.m[3][2], m.m[0][3], m.m[1][3],
                     m.m[2][3], m.m[3][3]);
}

// Gauss-Jordan with full pivoting, transform.cpp:82-136.
Matrix4x4 Inverse(const Matrix4x4 &m, bool *singular) {
    int indxc[4], indxr[4];
    int ipiv[4] = {0, 0, 0, 0};
    float minv[4][4];
    std::memcpy(minv, m.m, 4 * 4 * sizeof(float));
    if (singular) *singular = false;
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        float big = 0.f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] != 1) {
                for (int k = 0; k < 4; k++) {
                    if (ipiv[k] == 0) {
                        if (std::abs(minv[j][k]) >= big) {
                            big = float(std::abs(minv[j][k]));
                            irow = j;
                            icol = k;
                        }
                    } else if (ipiv[k] > 1) {
                        if (singular) *singular = true;
                    }
                }
            }
        }
        ++ipiv[icol];
        if (irow != icol) {
            for (int k = 0; k < 4; ++k) std::swap(minv[irow][k], minv[icol][k]);
        }
        indxr[i] = irow;
        indxc[i] = icol;
        if (minv[icol][icol] == 0.f) {
            if (singular) *singular = true;
        }
        // "Float pivinv = 1. / minv[icol][icol]" is a double division rounded to float
        float pivinv = (float)(1. / (double)minv[icol][icol]);
        minv[icol][icol] = 1.;
        for (int j = 0; j < 4; j++) minv[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j != icol) {
                float save = minv[j][icol];
                minv[j][icol] = 0;
                for (int k = 0; k < 4; k++) minv[j][k] -= minv[icol][k] * save;
            }
        }
    }
    for (int j = 3; j >= 0; j--) {
        if (indxr[j] != indxc[j]) {
            for (int k = 0; k < 4; k++) std::swap(minv[k][indxr[j]], minv[k][indxc[j]]);
        }
    }
    Matrix4x4 r;
    std::memcpy(r.m, minv, sizeof(minv));
    return r;
}

Bounds3 Transform::Bounds(const Bounds3 &b) const {
    Bounds3 ret(Point(Vec3(b.pMin.x, b.pMin.y, b.pMin.z)));
    ret = Union(ret, Point(Vec3(b.pMax.x, b.pMin.y, b.pMin.z)));
    ret = Union(ret, Point(Vec3(b.pMin.x, b.pMax.y, b.pMin.z)));
    ret = Union(ret, Point(Vec3(b.pMin.x, b.pMin.y, b.pMax.z)));
    ret = Union(ret, Point(Vec3(b.pMin.x, b.pMax.y, b.pMax.z)));
    ret = Union(ret, Point(Vec3(b.pMax.x, b.pMax.y, b.pMin.z)));
    ret = Union(ret, Point(Vec3(b.pMax.x, b.pMin.y, b.pMax.z)));
    ret = Union(ret, Point(Vec3(b.pMax.x, b.pMax.y, b.pMax.z)));
    return ret;
}

Transform Translate(const Vec3 &d) {  // transform.cpp:141-147
    Matrix4x4 m(1, 0, 0, d.x, 0, 1, 0, d.y, 0, 0, 1, d.z, 0, 0, 0, 1);
    Matrix4x4 minv(1, 0, 0, -d.x, 0, 1, 0, -d.y, 0, 0, 1, -d.z, 0, 0, 0, 1);
    return Transform(m, minv);
}

Transform Scale(float x, float y, float z) {  // transform.cpp:149-153
    Matrix4x4 m(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1);
    Matrix4x4 minv(1 / x, 0, 0, 0, 0, 1 / y, 0, 0, 0, 0, 1 / z, 0, 0, 0, 0, 1);
    return Transform(m, minv);
}

Transform Rotate(float theta, const Vec3 &axis) {  // transform.cpp:179-201
    Vec3 a = Normalize(axis);
    float sinTheta = std::sin(Radians(theta));
    float cosTheta = std::cos(Radians(theta));
    Matrix4x4 m;
    m.m[0][0] = a.x * a.x + (1 - a.x * a.x) * cosTheta;
    m.m[0][1] = a.x * a.y * (1 - cosTheta) - a.z * sinTheta;
    m.m[0][2] = a.x * a.z * (1 - cosTheta) + a.y * sinTheta;
    m.m[0][3] = 0;
    m.m[1][0] = a.x * a.y * (1 - cosTheta) + a.z * sinTheta;
    m.m[1][1] = a.y * a.y + (1 - a.y * a.y) * cosTheta;
    m.m[1][2] = a.y * a.z * (1 - cosTheta) - a.x * sinTheta;
    m.m[1][3] = 0;
    m.m[2][0] = a.x * a.z * (1 - cosTheta) - a.y * sinTheta;
    m.m[2][1] = a.y * a.z * (1 - cosTheta) + a.x * sinTheta;
    m.m[2][2] = a.z * a.z + (1 - a.z * a.z) * cosTheta;
    m.m[2][3] = 0;
    return Transform(m, Transpose(m));
}

Transform LookAt(const Vec3 &pos, const Vec3 &look, const Vec3 &up, bool *degenerate) {
    // transform.cpp:203-241
    Matrix4x4 c2w;
    c2w.m[0][3] = pos.x;
    c2w.m[1][3] = pos.y;
    c2w.m[2][3] = pos.z;
    c2w.m[3][3] = 1;
    Vec3 dir = Normalize(look - pos);
    if (degenerate) *degenerate = false;
    if (Cross(Normalize(up), dir).Length() == 0) {
        if (degenerate) *degenerate = true;
        return Transform();
    }
    Vec3 right = Normalize(Cross(Normalize(up), dir));
    Vec3 newUp = Cross(dir, right);
    c2w.m[0][0] = right.x; c2w.m[1][0] = right.y; c2w.m[2][0] = right.z; c2w.m[3][0] = 0.;
    c2w.m[0][1] = newUp.x; c2w.m[1][1] = newUp.y; c2w.m[2][1] = newUp.z; c2w.m[3][1] = 0.;
    c2w.m[0][2] = dir.x;   c2w.m[1][2] = dir.y;   c2w.m[2][2] = dir.z;   c2w.m[3][2] = 0.;
    return Transform(Inverse(c2w), c2w);
}

Transform Perspective(float fov, float n, float f) {  // transform.cpp:290-299
    Matrix4x4 persp(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, f / (f - n), -f * n / (f - n), 0, 0, 1, 0);
    float invTanAng = 1 / std::tan(Radians(fov) / 2);
    return Scale(invTanAng, invTanAng, 1) * Transform(persp);
}

}  // namespace mipt
