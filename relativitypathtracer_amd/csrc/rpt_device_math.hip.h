// rpt_device_math.hip.h — fp32 vector helpers for the gfx950 render kernels.
//
// Every helper spells out one IEEE-754 binary32 operation order; nothing here may be contracted
// into FMAs or re-associated (the translation unit is built with -ffp-contract=off and the pragma
// below pins it), and / and sqrt are the correctly rounded forms (hipcc default,
// -fhip-fp32-correctly-rounded-divide-sqrt), because results are compared bit for bit with the
// CPU oracle.  Built-in semantics follow OpenCL C 1.2 §6.12 as listed in oracle/rpt_oracle.c.
#pragma once
#include <hip/hip_runtime.h>

#ifdef RPT_RELAXED_FP
#pragma clang fp contract(fast)      /* rpt_relaxed.hip only: the opt-in "OpenCL-conformant arithmetic" build of the same source */
#else
#pragma clang fp contract(off)
#endif

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

// ---- three quotients by ONE scalar (normalize(), dir / scale: opencl_kernel.cl:216, 317, 339, 587-593) ---------------------------
// The correctly rounded x / s costs 11 instructions (v_div_scale x 2, v_rcp, four fmas, v_div_fmas, v_div_fixup), three of them 33.
// What the three have in common is the reciprocal: r = rcp(s) refined by one Newton step is the correctly rounded 1 / s for every
// s whose significand is not all ones (Markstein), and a quotient is then finished by residual corrections
//     q = n r;   e = fma(-s, q, n);   q = fma(e, r, q)          [ROUNDS times]
// in which every residual e is exact.  That is the tail of the hardware sequence without its scaling — valid while nothing
// overflows, underflows or is a zero (a signed zero comes out of the fmas with the wrong sign): the fast path is taken only for a
// wave whose every lane has 2^-40 <= |s| <= 2^40 and 2^-60 <= |n| <= 2^60 for all three numerators (NaNs fail the test), anything
// else takes the IEEE divisions.  Whether the result equals x / s bit for bit on that domain is not argued but TESTED on the
// device (rpt_probe_division: 10^9 quotients, all-ones significands, the actual normalize() inputs of a frame).
// RPT_SHARED_RCP = number of residual corrections (1 or 2); undefined or 0 = three IEEE divisions (the product as shipped).
template <int ROUNDS>
RPT_DEV f3 div3_shared_unguarded(f3 a, float s) {
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float r = __builtin_fmaf(__builtin_fmaf(-s, r0, 1.0f), r0, r0);
    float qx = a.x * r, qy = a.y * r, qz = a.z * r;
#pragma unroll
    for (int k = 0; k < ROUNDS; k++) {
        qx = __builtin_fmaf(__builtin_fmaf(-s, qx, a.x), r, qx);
        qy = __builtin_fmaf(__builtin_fmaf(-s, qy, a.y), r, qy);
        qz = __builtin_fmaf(__builtin_fmaf(-s, qz, a.z), r, qz);
    }
    return mk3(qx, qy, qz);
}
RPT_DEV bool div3_shared_domain(f3 a, float s) {
    const float as = __builtin_fabsf(s), ax = __builtin_fabsf(a.x), ay = __builtin_fabsf(a.y), az = __builtin_fabsf(a.z);
    return (as >= 0x1p-40f) & (as <= 0x1p40f) & (ax >= 0x1p-60f) & (ax <= 0x1p60f) & (ay >= 0x1p-60f) & (ay <= 0x1p60f) & (az >= 0x1p-60f) & (az <= 0x1p60f);
}
template <int ROUNDS>
RPT_DEV f3 div3_shared(f3 a, float s) {
    if (__builtin_amdgcn_ballot_w64(!div3_shared_domain(a, s)) == 0ull) return div3_shared_unguarded<ROUNDS>(a, s);      // wave-uniform
    return mk3(a.x / s, a.y / s, a.z / s);
}
#if defined(RPT_SHARED_RCP) && RPT_SHARED_RCP > 0
RPT_DEV f3 operator/(f3 a, float s) { return div3_shared<RPT_SHARED_RCP>(a, s); }
#else
RPT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
#endif
RPT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RPT_DEV f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
RPT_DEV f4 operator*(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }

RPT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RPT_DEV float dot(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
RPT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RPT_DEV float length(f3 v) { return __builtin_sqrtf(dot(v, v)); }
RPT_DEV f3 normalize(f3 v) { const float l = length(v); return v / l; }      // (three divisions by one scalar: operator/ above)

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


// asin and atan2 of the textured-sphere (u,v) (opencl_kernel.cl:356-357).  OpenCL leaves their last bits to the
// implementation (<= 4 ulp), so no particular rounding is "the reference's"; what matters here is that the oracle and
// this kernel agree bit for bit.  Both therefore carry the SAME explicit algorithm (the fdlibm single-precision
// kernels: argument reduction + minimax polynomial, every operation an IEEE fp32/fp64 +,-,*,/ or sqrt, no
// contraction) instead of their platforms' libm.  Accuracy: asin < 1 ulp, atan2 < 2 ulp (tests/test_oracle.py).
RPT_DEV float rpt_asinf(float x) {
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    const double pio2 = 1.570796326794896558e+00;
    const int hx = __float_as_int(x);
    const int ix = hx & 0x7fffffff;
    if (ix >= 0x3f800000) {                        // |x| >= 1
        if (ix == 0x3f800000) return (float)(x * pio2);
        return (x - x) / (x - x);                  // NaN
    }
    if (ix < 0x3f000000) {                         // |x| < 0.5
        if (ix < 0x39800000) return x;             // |x| < 2^-12
        const float t = x * x;
        const float p = t * (pS0 + t * (pS1 + t * pS2));
        const float q = 1.0f + t * qS1;
        const float w = p / q;
        return x + x * w;
    }
    const float w0 = 1.0f - __builtin_fabsf(x);
    const float t = w0 * 0.5f;
    const float p = t * (pS0 + t * (pS1 + t * pS2));
    const float q = 1.0f + t * qS1;
    const double s = __builtin_sqrt((double)t);
    const float w = p / q;
    const float r = (float)(pio2 - 2.0 * (s + s * (double)w));
    return hx > 0 ? r : -r;
}

RPT_DEV float rpt_atanf(float x) {
    const float aT0 = 3.3333328366e-01f, aT1 = -1.9999158382e-01f, aT2 = 1.4253635705e-01f, aT3 = -1.0648017377e-01f, aT4 = 6.1687607318e-02f;
    const int hx = __float_as_int(x);
    const int ix = hx & 0x7fffffff;
    if (ix >= 0x4c800000) {                        // |x| >= 2^26
        if (ix > 0x7f800000) return x + x;         // NaN
        const float r = 1.5707962513e+00f + 7.5497894159e-08f;
        return hx > 0 ? r : -r;
    }
    int id;
    float hi = 0.0f, lo = 0.0f;
    if (ix < 0x3ee00000) {                         // |x| < 0.4375
        if (ix < 0x39800000) return x;             // |x| < 2^-12
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {                     // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; hi = 4.6364760399e-01f; lo = 5.0121582440e-09f; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; hi = 7.8539812565e-01f; lo = 3.7748947079e-08f; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; hi = 9.8279368877e-01f; lo = 3.4473217170e-08f; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; hi = 1.5707962513e+00f; lo = 7.5497894159e-08f; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * aT4));
    const float s2 = w * (aT1 + w * aT3);
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

RPT_DEV float rpt_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int hx = __float_as_int(x), hy = __float_as_int(y);
    const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;      // NaN
    if (hx == 0x3f800000) return rpt_atanf(y);                 // x = 1
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);               // 2*sign(x) + sign(y)
    if (iy == 0) return m == 0 || m == 1 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 26) { z = pi_o_2 + 0.5f * pi_lo; m &= 1; }         // |y/x| > 2^26
    else if (k < -26 && hx < 0) z = 0.0f;                      // 0 > |y|/x > -2^-26
    else z = rpt_atanf(__builtin_fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

}  // namespace rptd
