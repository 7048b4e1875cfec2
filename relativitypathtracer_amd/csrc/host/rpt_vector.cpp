// rpt_vector.cpp — fp32 vector / matrix / Lorentz-boost helpers of the host side.
//
// Restates the arithmetic of the reference's Vector.cpp (cited per function) so that the
// Object[] bytes handed to the render path are the ones the reference host would produce.
// "Next" row f1 of SURVEY.md §8(f).  Built with -ffp-contract=off.
#include "rpt_vector.h"

#include <cmath>
#include <cstring>

namespace rpt {

// Vector.cpp:4-15
float sqr_magnitude(const rpt_float3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
float magnitude(const rpt_float3 v) { return std::sqrt(sqr_magnitude(v)); }
rpt_float3 normalize(const rpt_float3 v) {
    const float m = magnitude(v);
    return make_float3(v.x / m, v.y / m, v.z / m);
}

// Vector.cpp:17-63 — all 3-component; the result's .w is always 0
rpt_float3 operator+(const rpt_float3 &a, const rpt_float3 &b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
rpt_float3 &operator+=(rpt_float3 &a, const rpt_float3 &b) { a = a + b; return a; }
rpt_float3 operator-(const rpt_float3 &a, const rpt_float3 &b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
rpt_float3 operator-(const rpt_float3 &v) { return make_float3(-v.x, -v.y, -v.z); }
rpt_float3 operator*(const rpt_float3 &v, const float &c) { return make_float3(v.x * c, v.y * c, v.z * c); }
rpt_float3 operator*(const float &c, const rpt_float3 &v) { return v * c; }
rpt_float3 operator/(const rpt_float3 &v, const float &c) { return make_float3(v.x / c, v.y / c, v.z / c); }

// Vector.cpp:65-67 — 4-component
float dot(const rpt_float4 &a, const rpt_float4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Vector.cpp:69-75
rpt_float3 cross(const rpt_float3 &a, const rpt_float3 &b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Vector.cpp:77-91 — Windows min/max macros: (a < b) ? a : b and (a > b) ? a : b
rpt_float3 elementwise_min(const rpt_float3 &a, const rpt_float3 &b) {
    return make_float3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z);
}
rpt_float3 elementwise_max(const rpt_float3 &a, const rpt_float3 &b) {
    return make_float3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z);
}

// Vector.cpp:94-149 — inverse by cofactors.  Entry (i,j) of the inverse is
// (-1)^(i+j) / det * minor3(M without row j and column i), each 3x3 minor expanded along its first
// row with 2x2 minors of its last two rows — the same products, in the same order, as the
// reference's hand-unrolled form.
bool calcInvM(rpt_object &object) {
    float m[4][4];
    std::memcpy(m, object.M, sizeof m);
    auto minor2 = [&](int r0, int r1, int c0, int c1) { return m[r0][c0] * m[r1][c1] - m[r0][c1] * m[r1][c0]; };
    auto minor3 = [&](int skip_row, int skip_col) {
        int r[3], c[3];
        for (int k = 0, n = 0; k < 4; k++) if (k != skip_row) r[n++] = k;
        for (int k = 0, n = 0; k < 4; k++) if (k != skip_col) c[n++] = k;
        return m[r[0]][c[0]] * minor2(r[1], r[2], c[1], c[2]) - m[r[0]][c[1]] * minor2(r[1], r[2], c[0], c[2]) +
               m[r[0]][c[2]] * minor2(r[1], r[2], c[0], c[1]);
    };
    float det = m[0][0] * minor3(0, 0) - m[0][1] * minor3(0, 1) + m[0][2] * minor3(0, 2) - m[0][3] * minor3(0, 3);
    if (det == 0.0f) return false;
    det = 1 / det;
    float inv[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            const float cof = minor3(j, i);
            inv[i][j] = det * (((i + j) & 1) ? -cof : cof);
        }
    std::memcpy(object.InvM, inv, sizeof inv);
    return true;
}

// Vector.cpp:151-166 — axis/angle rotation, column-scaled, translation in the 4th column
void TRS(rpt_object &object, rpt_float3 translation, float angle, rpt_float3 axis, rpt_float3 scale) {
    rpt_float3 R[3] = {make_float3(1, 0, 0), make_float3(0, 1, 0), make_float3(0, 0, 1)};
    if (angle != 0) {
        const float c = std::cos(angle);
        const float s = std::sin(angle);
        const rpt_float3 u = normalize(axis);
        R[0] = make_float3(c + u.x * u.x * (1 - c), u.x * u.y * (1 - c) - u.z * s, u.x * u.z * (1 - c) + u.y * s);
        R[1] = make_float3(u.y * u.x * (1 - c) + u.z * s, c + u.y * u.y * (1 - c), u.y * u.z * (1 - c) - u.x * s);
        R[2] = make_float3(u.z * u.x * (1 - c) - u.y * s, u.z * u.y * (1 - c) + u.x * s, c + u.z * u.z * (1 - c));
    }
    const float t[3] = {translation.x, translation.y, translation.z};
    for (int r = 0; r < 3; r++) object.M[r] = make_float4(R[r].x * scale.x, R[r].y * scale.y, R[r].z * scale.z, t[r]);
    object.M[3] = make_float4(0, 0, 0, 1);
    calcInvM(object);
}

// Vector.cpp:168-173
void Identity(rpt_float4 (&M)[4]) {
    for (int r = 0; r < 4; r++) M[r] = make_float4(r == 0, r == 1, r == 2, r == 3);
}

// Vector.cpp:175-187 — boost matrix for velocity v (units of c), rows/cols ordered (t,x,y,z)
void Lorentz(rpt_float4 (&M)[4], rpt_float3 v) {
    const float gamma = 1.0f / std::sqrt(1.0f - dot(v, v));
    const float vSqr = dot(v, v);
    if (vSqr == 0) {
        Identity(M);
        return;
    }
    const float c[3] = {v.x, v.y, v.z};
    M[0] = make_float4(gamma, -v.x * gamma, -v.y * gamma, -v.z * gamma);
    for (int i = 0; i < 3; i++) {
        float s[3];
        for (int j = 0; j < 3; j++) {
            s[j] = (gamma - 1.0f) * c[i] * c[j] / vSqr;
            if (i == j) s[j] = s[j] + 1.0f;
        }
        M[i + 1] = make_float4(-c[i] * gamma, s[0], s[1], s[2]);
    }
}

// Vector.cpp:189-193 — relativistic velocity addition; the gamma/(1+gamma) factor is evaluated in
// double (the reference writes the literal 1.0) and rounded to float when it multiplies the vector
rpt_float3 AddVelocity(rpt_float3 const &v1, rpt_float3 const &v2) {
    const float gamma_v = 1.0f / std::sqrt(1 - dot(v1, v1));
    const float k = (float)(gamma_v / (1.0 + gamma_v));
    return 1.0f / (1.0f + dot(v2, v1)) * (v1 + v2 + k * cross(v1, cross(v1, v2)));
}

static inline rpt_float4 column(rpt_float4 const (&B)[4], int j) {
    const float *b0 = &B[0].x, *b1 = &B[1].x, *b2 = &B[2].x, *b3 = &B[3].x;
    return make_float4(b0[j], b1[j], b2[j], b3[j]);
}

// Vector.cpp:195-204 — A <- A * B (row i is fully replaced before row i+1 is read; rows are independent)
void MatrixMultiplyLeft(rpt_float4 (&A)[4], rpt_float4 const (&B)[4]) {
    for (int i = 0; i < 4; i++) {
        const rpt_float4 a = A[i];
        A[i] = make_float4(dot(a, column(B, 0)), dot(a, column(B, 1)), dot(a, column(B, 2)), dot(a, column(B, 3)));
    }
}

// Vector.cpp:206-220 — B <- A * B
void MatrixMultiplyRight(rpt_float4 const (&A)[4], rpt_float4 (&B)[4]) {
    rpt_float4 out[4];
    for (int i = 0; i < 4; i++)
        out[i] = make_float4(dot(A[i], column(B, 0)), dot(A[i], column(B, 1)), dot(A[i], column(B, 2)), dot(A[i], column(B, 3)));
    for (int i = 0; i < 4; i++) B[i] = out[i];
}

// Vector.cpp:222-232 — Lorentz = boost(v); InvLorentz = boost(-v) built by flipping the time column
void setLorentzBoost(rpt_object &object, rpt_float3 v) {
    Lorentz(object.Lorentz, v);
    const float gamma = 1.0f / std::sqrt(1.0f - dot(v, v));
    object.InvLorentz[0] = make_float4(gamma, v.x * gamma, v.y * gamma, v.z * gamma);
    for (int r = 1; r < 4; r++) {
        object.InvLorentz[r] = object.Lorentz[r];
        object.InvLorentz[r].x *= -1;
    }
}

rpt_object defaultObject() {
    rpt_object o;
    std::memset(&o, 0, sizeof o);
    Identity(o.Lorentz);
    Identity(o.InvLorentz);
    o.textureIndex = -1;
    return o;
}

}  // namespace rpt
