// ptmath.cpp -- the 4x4 algebra behind the .pbrt transform directives.
//
// What has to agree with the reference is the *value* of every CTM, bit for bit: world-space
// vertices, the camera matrices and the BVH all follow from them (tests/test_frontend.py pins
// the killeroo scene statistics, tests/test_oracle_pins.py the camera rays). For float
// arithmetic that means the same operations on the same operands in the same order as
// src/core/transform.cpp -- which pivot an inversion picks, which product is rounded first in
// a rotation entry -- and that is the only thing taken from there. The code itself is this
// repository's: a flat 16-float working copy, a pivot bitmask and a list of recorded
// exchanges for the inversion; the rotation assembled from its diagonal / off-diagonal
// Rodrigues terms; frames built column by column.
#include "ptmath.h"

namespace mipt {

Matrix4x4 Transpose(const Matrix4x4 &m) {
    Matrix4x4 t;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) t.m[c][r] = m.m[r][c];
    return t;
}

namespace {

// Working copy of a 4x4 during elimination: element (r, c) at 4r + c.
struct Grid {
    float a[16];
    float &at(int r, int c) { return a[4 * r + c]; }
    void ExchangeRows(int r0, int r1) {
        if (r0 != r1)
            for (int c = 0; c < 4; ++c) std::swap(at(r0, c), at(r1, c));
    }
    void ExchangeColumns(int c0, int c1) {
        if (c0 != c1)
            for (int r = 0; r < 4; ++r) std::swap(at(r, c0), at(r, c1));
    }
};

struct Exchange { int row, col; };

}  // namespace

// Gauss-Jordan elimination with full pivoting (the reference's choice of method, transform.cpp:82-136).
// Order-sensitive details kept: the pivot search scans rows then columns over the not-yet-reduced
// part and takes the *last* of equal maxima (>=); the pivot's reciprocal is a double division
// rounded to float ("1. / x" in the reference); the pivot row is scaled before the other rows are
// reduced with `x -= pivotRow * factor`; the row exchanges are undone as column exchanges, newest first.
Matrix4x4 Inverse(const Matrix4x4 &m, bool *singular) {
    Grid g;
    std::memcpy(g.a, m.m, sizeof(g.a));
    bool isSingular = false;
    unsigned reduced = 0;   // bit c: column c already holds a pivot
    Exchange done[4];
    for (int step = 0; step < 4; ++step) {
        int pr = 0, pc = 0;
        float best = 0.f;
        for (int r = 0; r < 4; ++r) {
            if ((reduced >> r) & 1u) continue;
            for (int c = 0; c < 4; ++c) {
                if ((reduced >> c) & 1u) continue;
                const float mag = std::abs(g.at(r, c));
                if (mag >= best) { best = mag; pr = r; pc = c; }
            }
        }
        // (a column can be picked twice only for a singular matrix: `reduced` then already has its bit)
        if ((reduced >> pc) & 1u) isSingular = true;
        reduced |= 1u << pc;
        g.ExchangeRows(pr, pc);   // the pivot moves onto the diagonal
        done[step] = Exchange{pr, pc};
        float &pivot = g.at(pc, pc);
        if (pivot == 0.f) isSingular = true;
        const float scale = (float)(1. / (double)pivot);
        pivot = 1.f;
        for (int c = 0; c < 4; ++c) g.at(pc, c) *= scale;
        for (int r = 0; r < 4; ++r) {
            if (r == pc) continue;
            const float factor = g.at(r, pc);
            g.at(r, pc) = 0.f;
            for (int c = 0; c < 4; ++c) g.at(r, c) -= g.at(pc, c) * factor;
        }
    }
    for (int step = 3; step >= 0; --step) g.ExchangeColumns(done[step].row, done[step].col);
    if (singular) *singular = isSingular;
    Matrix4x4 out;
    std::memcpy(out.m, g.a, sizeof(g.a));
    return out;
}

// Box of the eight transformed corners (transform.cpp:243-254; min / max are exact, so the order of the corners is free).
Bounds3 Transform::Bounds(const Bounds3 &b) const {
    const Vec3 lo = b.pMin, hi = b.pMax;
    Bounds3 out(Point(lo));
    for (int corner = 1; corner < 8; ++corner)
        out = Union(out, Point(Vec3((corner & 1) ? hi.x : lo.x, (corner & 2) ? hi.y : lo.y, (corner & 4) ? hi.z : lo.z)));
    return out;
}

Transform Translate(const Vec3 &d) {  // transform.cpp:141-147
    Matrix4x4 fwd, back;   // identity
    const float t[3] = {d.x, d.y, d.z};
    for (int r = 0; r < 3; ++r) { fwd.m[r][3] = t[r]; back.m[r][3] = -t[r]; }
    return Transform(fwd, back);
}

Transform Scale(float x, float y, float z) {  // transform.cpp:149-153
    Matrix4x4 fwd, back;
    const float s[3] = {x, y, z};
    for (int r = 0; r < 3; ++r) { fwd.m[r][r] = s[r]; back.m[r][r] = 1 / s[r]; }
    return Transform(fwd, back);
}

// Rotation by theta degrees about an axis (transform.cpp:179-201). Entry (i, i) is a_i^2 + (1 - a_i^2) cos;
// entry (i, j), i != j, is a_i a_j (1 - cos) -/+ a_k sin with k the third axis and the sign of the
// permutation (i, j, k): each written so that its products round in the reference's order.
Transform Rotate(float theta, const Vec3 &axis) {
    const Vec3 u = Normalize(axis);
    const float a[3] = {u.x, u.y, u.z};
    const float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 rot;   // identity: row / column 3 stay (0, 0, 0, 1)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            if (i == j) { rot.m[i][i] = a[i] * a[i] + (1 - a[i] * a[i]) * c; continue; }
            const int k = 3 - i - j;
            const bool even = (j == (i + 1) % 3);   // (i, j, k) is a cyclic permutation of (0, 1, 2)
            const int lo = i < j ? i : j, hi = i < j ? j : i;   // the reference writes the product as a_lo * a_hi
            const float sym = a[lo] * a[hi] * (1 - c), skew = a[k] * s;
            rot.m[i][j] = even ? sym - skew : sym + skew;
        }
    return Transform(rot, Transpose(rot));
}

// Camera frame from an eye point, a target and an up hint (transform.cpp:203-241): columns right / up / forward /
// position of camera-to-world; the returned transform is world-to-camera.
Transform LookAt(const Vec3 &pos, const Vec3 &look, const Vec3 &up, bool *degenerate) {
    const Vec3 forward = Normalize(look - pos);
    const Vec3 side = Cross(Normalize(up), forward);
    if (degenerate) *degenerate = false;
    if (side.Length() == 0) {   // up and viewing direction are parallel
        if (degenerate) *degenerate = true;
        return Transform();
    }
    const Vec3 right = Normalize(side);
    const Vec3 newUp = Cross(forward, right);
    const Vec3 cols[4] = {right, newUp, forward, pos};
    Matrix4x4 camToWorld;
    for (int c = 0; c < 4; ++c) {
        camToWorld.m[0][c] = cols[c].x;
        camToWorld.m[1][c] = cols[c].y;
        camToWorld.m[2][c] = cols[c].z;
        camToWorld.m[3][c] = (c == 3) ? 1.f : 0.f;
    }
    return Transform(Inverse(camToWorld), camToWorld);
}

// Perspective projection onto z in [0, 1] followed by the field-of-view scale (transform.cpp:290-299).
Transform Perspective(float fov, float n, float f) {
    Matrix4x4 proj;
    proj.m[2][2] = f / (f - n);
    proj.m[2][3] = -f * n / (f - n);
    proj.m[3][2] = 1;
    proj.m[3][3] = 0;
    const float invTan = 1 / std::tan(Radians(fov) / 2);
    return Scale(invTan, invTan, 1) * Transform(proj);
}

}  // namespace mipt
