// rpt_device_math.hip.h — fp32 vector helpers for the gfx950 render kernels.
//
// Every helper spells out one IEEE-754 binary32 operation order; nothing here may be contracted
// into FMAs or re-associated (the translation unit is built with -ffp-contract=off and the pragma
// below pins it), and / and sqrt are the correctly rounded forms (hipcc default,
// -fhip-fp32-correctly-rounded-divide-sqrt), because results are compared bit for bit with the
// CPU oracle.  Built-in semantics follow OpenCL C 1.2 §6.12 as listed in oracle/rpt_oracle.c.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

#define RPT_DEV __device__ __forceinline__

namespace rptd {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

RPT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RPT_DEV f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
RPT_DEV f3 yzw(f4 v) { return mk3(v.y, v.z, v.w); }
RPT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RPT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RPT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RPT_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
RPT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RPT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RPT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RPT_DEV f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
RPT_DEV f4 operator*(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }

RPT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RPT_DEV float dot(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
RPT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RPT_DEV float length(f3 v) { return __builtin_sqrtf(dot(v, v)); }
RPT_DEV f3 normalize(f3 v) { const float l = length(v); return mk3(v.x / l, v.y / l, v.z / l); }

RPT_DEV float cl_min(float x, float y) { return y < x ? y : x; }     // OpenCL min(): y < x ? y : x
RPT_DEV float cl_max(float x, float y) { return x < y ? y : x; }     // OpenCL max(): x < y ? y : x
RPT_DEV int imin(int x, int y) { return y < x ? y : x; }
RPT_DEV int iclamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
RPT_DEV float cl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f)); }

// saturating float -> int, NaN -> 0 (same definition as the oracle's f2i_sat)
RPT_DEV int f2i_sat(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

// exact fmod(x, 0.5f): x - 0.5*trunc(2x) is computed without rounding for every finite x
// (2x, trunc, the product and the difference are all exactly representable); zero results keep the
// sign of x and non-finite x gives NaN, as C fmodf does.
RPT_DEV float fmod_half(float x) {
    const float r = x - 0.5f * __builtin_truncf(x * 2.0f);
    return r == 0.0f ? __builtin_copysignf(0.0f, x) : r;
}

// row-major 4x4 (rows of rpt_float4) times vector, as transformPoint/transformPoint4D/
// transformDirection/applyTranspose of the reference (opencl_kernel.cl:75-104)
struct Mat4 { f4 r[4]; };

RPT_DEV f4 ld4(const rpt_float4 &v) { return mk4(v.x, v.y, v.z, v.w); }
RPT_DEV f3 ld3(const rpt_float4 &v) { return mk3(v.x, v.y, v.z); }

RPT_DEV f3 transformPoint(const rpt_float4 *M, f3 v) {
    const f4 V = mk4(v.x, v.y, v.z, 1.0f);
    return mk3(dot(ld4(M[0]), V), dot(ld4(M[1]), V), dot(ld4(M[2]), V));
}
RPT_DEV f4 transformPoint4D(const rpt_float4 *M, f4 v) {
    return mk4(dot(ld4(M[0]), v), dot(ld4(M[1]), v), dot(ld4(M[2]), v), dot(ld4(M[3]), v));
}
RPT_DEV f3 transformDirection(const rpt_float4 *M, f3 v) {
    return mk3(dot(ld3(M[0]), v), dot(ld3(M[1]), v), dot(ld3(M[2]), v));
}
RPT_DEV f3 applyTranspose(const rpt_float4 *M, f3 v) {
    return ld3(M[0]) * v.x + ld3(M[1]) * v.y + ld3(M[2]) * v.z;
}

}  // namespace rptd
