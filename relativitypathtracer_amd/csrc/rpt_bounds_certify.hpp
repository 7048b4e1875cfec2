// rpt_bounds_certify.hpp — the PROOF behind the in-wave cull.  rpt_screen_bounds.hpp proposes, per object and frame, a region of
// the image plane (a rectangle, optionally cut to an octagon) outside which no primary ray is supposed to reach the object; how it
// finds that region (adaptive sampling of an outline, margins, a horizon rule ...) is heuristic, and three rounds of soak runs kept
// finding outlines the sampling had not followed.  Nothing below depends on how the region was found: certify() takes the region
// as a CLAIM and proves, in outward-rounded arithmetic, that the kernel's own float computation cannot report a hit of the object
// for any pixel of the frame outside it — or fails, and the object then gets the full plane (it is tested for every pixel, as in the
// reference, opencl_kernel.cl:382-425).  A cull is therefore never applied on the strength of sampling.
//
// ---------------------------------------------------------------------------------------------------------------------------
// 1. What the kernel computes (rpt_kernels.hip.h: render_pixel_body, createCamRayDir, intersect_object_primary; fp32, no
//    contraction; u = 2^-24 is the unit roundoff, every bound below is first order in u with the constant rounded up)
//
//    pixel -> (fu, fv) floats, a point of the WINDOW  |fu| <= 2, |fv| <= 1/2  when the frame is at most 4 : 1 (launch() culls only
//      then).  Real numbers from here on: whatever rounding produced them, they are SOME point of the window, and every point of the
//      window outside the claimed region is covered by the proof.
//    nd   = normalize(normalize((fu, fv, 1/2)))   TWICE, as the reference does (createCamRay normalises, intersect_scene normalises again:
//                                              opencl_kernel.cl:71, 387; rpt_kernels.hip.h: createCamRayDir, trace).  One normalisation:
//                                              3-term dot 3u, sqrt 1.5u + u, divide u = 3.5u per component; the second one works on a
//                                              vector whose own length is 1 +- 3.5u:   nd_j = d_j / |d| (1 + e), |e| <= 10.6u
//    d3_k = dot(Lorentz[k], (interval, nd))   4-term dot: error <= 4.1u sum_j |L_kj| |w_j|; with the input error of nd:
//                                              |d3_k - (L w)_k| <= 14.7u sum_j |L_kj|           (|w_j| <= 1)
//    dir  = transformDirection(InvM, d3)      3-term dot: |dir_i - (I L w)_i| <= 17.8u sum_k |I_ik| sum_j |L_kj|
//    dirn = dir / length(dir)                 dirn_i = dir_i / |dir| (1 + e), |e| <= 3.5u: as a DIRECTION, dir_i (1 + e)
//    => the direction the intersector gets is, up to a positive factor,   A w + e,   A = InvM3 * Lorentz[1..3][0..3] (exact reals
//       formed from the float matrices), w = (interval, nd), and   |e_i| <= ERR_i := 24u * sum_k |I_ik| sum_j |L_kj|   (17.8u + 3.5u of
//       |dir_i| <= that same sum = 21.4u, rounded up).
//    origin: DObj.ox/oy/oz — floats computed ONCE per frame on the host (build_dobjs) and read by the kernel: the certificate uses
//       the very same floats, so the origin carries no error at all.
//
// 2. What a float "hit" implies about the EXACT ray (origin o, direction g = the float vector dirn), per intersector:
//    sphere (sphere_core, opencl_kernel.cl:335-359): hit => disc = fl(fl(b*b) - c) >= 0 with b = fl(dot(-o, g)), c = fl(fl(|o|^2) - 1).
//       |b - (-o . g)| <= 3u |o| |g|; |fl(b*b) - b^2| <= u b^2; |c - (|o|^2 - 1)| <= 4u |o|^2 + u; the last subtraction u |disc|; and
//       | |g|^2 - 1 | <= 8u.  Together the exact line's distance from the centre obeys  dist^2 <= 1 + 30u |o|^2 + 2u.  A hit also needs
//       fl(b + sqrt(disc)) > 1e-7, which for an origin outside the sphere (c > 0: sqrt(fl(b*b) - c) <= |b| in IEEE arithmetic) forces
//       b > 0: the FORWARD ray passes within R' of the centre,        R'^2 = 1 + 32u (|o|^2 + 1)        (32u = 2^-19).
//    cube (cube_core, :312-333): a hit through face U means d = fl(fl(+-1 - o_U) / g_U) >= 0 and |fl(o_V + fl(g_V d))| < 1 (same W).
//       The exact point P = o + g d (d >= 0: on the forward ray) has |P_U -+ 1| <= 2.1u (1 + |o_U|) and |P_V| < 1 + 2.1u + 1.1u |o_V|:
//       P lies in the cube [-m, m]^3,                                  m = 1 + 4u (1 + max_k |o_k|)     (4u = 2^-22).
//    mesh (octree_walk / intersect_octree, :200-308): no hit is reported unless intersect_AABB(root box) is true.  Its six plane
//       distances are t = fl(fl(b - o_k) * fl(1 / g_k)) = t_exact (1 + e), |e| <= 3.1u, the slab logic then guarantees a float
//       T = min of the far distances with T > 0 and T >= every near distance.  The exact point o + g T (forward ray) lies, per axis,
//       within 3.1u |b - o_k| of the slab [min_k, max_k] (infinite or NaN distances — g_k zero or denormal — only occur with o_k
//       itself inside the closed slab):   box grown per axis by          grow_k = 4u (max(|min_k|, |max_k|) + |o_k|) + 1e-30.
//    In every case: float hit  =>  the exact forward ray from o along the float direction meets the convex set B' given above.
//
// 3. The certificate.  Let K' be the convex cone of directions from o that meet B' (o outside B', checked).  With section 1, a float
//    hit at a point q of the window means  A w(q) + e  in K'  for some |e_i| <= ERR_i.  Multiply by |d| > 0: with W(q) = (interval |d|,
//    fu, fv, 1/2) — LINEAR in (s, fu, fv), s = interval |d| — a hit means  A W(q) + e in K', |e_i| <= |d| ERR_i.
//    SEGMENT TEST.  For a straight segment [qa, qb] of the image plane and s between interval * dmin and interval * dmax (the exact
//    range of |d| on the segment), the vectors A W lie in the convex hull of the four generators G = A (s, q, 1/2), s in {both ends},
//    q in {qa, qb}.  If a vector n satisfies       n . G + 2 dmax sum_i |n_i| ERR_i < 0      for all four generators, and
//    n . (p - o) > 0 for every p in B', then for every q of the segment and every error e with |e_i| <= 2 |d| ERR_i the ray from o
//    along A W(q) + e stays strictly on the negative side of the plane through o with normal n while B' is on the positive side:
//    no hit, with the error budget doubled (the factor 2 is what the connectedness argument below needs).  n is FOUND by a few steps
//    of the GJK minimum-norm iteration and then CHECKED by exactly those inequalities (with a relative slack of 1e-11 for the double
//    arithmetic, whose own error is 1e-15): only the check carries the proof.  A segment that cannot be certified is halved.
//    FROM THE BOUNDARY TO THE REGION.  Let P = window minus the claimed region (a closed polygonal set) and let every segment of the
//    boundary of P be certified.  Put D(nd) = A (interval, nd) for unit nd: the surface E of an ellipsoid, which contains the ball of
//    radius r0 = (1 - |c|) / ||A3^-1||_F about 0 (A3 c is its centre, 1 / ||A3^-1|| bounds its smallest half axis from below), where
//    A3 = InvM3 * Lorentz[1..3][1..3] and c = A3^-1 A[.][0] (|c| < 1 is checked: it
//    says the boosted null directions still cover the whole sphere, i.e. nd -> direction of D(nd) is a homeomorphism of the sphere).
//    The exact hit set H0 = { nd : D(nd) in K' } is connected (K' is a convex cone, its directions a connected set, the map a
//    homeomorphism).  The float hit set lies in H1 = { nd : D(nd) + e in K' for some |e_i| <= ERR_i }.  Take such an nd, D(nd) = k - e
//    with k in K'.  Walk y_t = k - (1 - t) e from D(nd) to k and project radially onto E: z_t = lambda_t y_t.  The points y_t stay
//    within |e| of a point of E, and a convex body that contains the ball of radius r0 has lambda_t <= 1 / (1 - |e| / r0) <= 4/3 when
//    |e| <= r0 / 4 (checked: "noise"; in coordinates scaled by 1 / ERR_i, in which |e| <= sqrt 3 — nothing here depends on the
//    coordinates being the object's own).  Hence z_t = lambda_t k - lambda_t (1 - t) e with lambda_t k in K' and an error of at most
//    4/3 ERR: the whole path lies in H2 = { nd : D(nd) + e in K', |e_i| <= 2 ERR_i } and ends in H0.  So the union of H0 and all
//    these paths is a CONNECTED set X with  H1 ⊂ X ⊂ H2.  The segment tests say that the boundary of P contains no point of H2, hence
//    none of X; a connected set that avoids the boundary of P lies inside P or outside it; and ONE witness direction — the ray to the
//    shape's centre, pushed through D in double and checked to hit the shape's inner half — is a point of H0 that is shown to lie
//    outside P (behind the camera, outside the window, or inside the claimed region with a margin).  Therefore X, and with it every
//    float hit, lies outside P: no pixel of the frame outside the claimed region can hit the object.                          q.e.d.
//
// What the proof does NOT cover is stated where it matters: it is a statement about the three intersectors as written in
// rpt_kernels.hip.h and compiled with IEEE division and square root and without contraction (a change to their arithmetic needs
// section 2 redone; the opt-in relaxed-arithmetic build, variants 50 / 51 — not a parity path — runs the same culls with a 2.5-ulp
// division inside budgets that were observed to be used to a fifth, which is an observation, not part of this proof), for frames of
// at most 4 : 1 and at most 2^20 pixels a side (launch() falls back to the un-culled kernel otherwise), and for finite matrices below
// 1e15 in magnitude (no float overflow).
#pragma once
#include <algorithm>
#include <cmath>

#include "../../include/rpt_layout.h"
#include "rpt_screen_bounds.hpp"

namespace rptb {
namespace cert {

constexpr double U24 = 5.9604644775390625e-8;        // 2^-24, unit roundoff of fp32
constexpr double WINDOW_U = 2.0, WINDOW_V = 0.5;      // every pixel of a frame of at most 4 : 1 (launch() culls only such frames)
constexpr int MAX_TESTS = 320;                        // segment tests per object before the object gets the full plane
constexpr int MAX_DEPTH = 14;                         // halvings of one boundary segment
constexpr int GJK_STEPS = 16;

enum Reason {
    CERTIFIED = 0,
    R_NONFINITE = 1,       // a matrix entry, the origin or a bound is not finite / too large for the error model
    R_NOT_HOMEOMORPHIC = 2,// |c| >= 1: the boosted directions do not cover the sphere once
    R_NOISE = 3,           // the float noise of the direction is not small against the smallest |D(nd)|
    R_ORIGIN_NEAR = 4,     // the ray origin is inside or too close to the inflated shape
    R_WITNESS = 5,         // no witness direction, or the witness lies in P: the claim is wrong (or cannot be told from wrong)
    R_BUDGET = 6,          // a boundary segment was still not separable at MAX_DEPTH, or MAX_TESTS ran out
    R_HIT_ON_BOUNDARY = 7  // an exact ray through a boundary point meets B': the claim is wrong or has no margin there
};

struct Stats { int reason, tests, max_depth, segments; };

struct Problem {
    double A[3][4];          // D = A (s, fu, fv, 1/2), s = interval |d|
    double err[3];           // ERR_i of section 1 (for unit nd)
    double o[3];             // the kernel's float origin, exactly
    int interval;
    int kind;                // 0 sphere, 1 box
    double R2;               // sphere: R'^2
    double lo[3], hi[3];     // box: B'
    double centre[3];        // of the shape
    // bookkeeping
    int tests = 0, max_depth = 0, segments = 0, reason = CERTIFIED;
};

inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// the kernel's float origin in object space, operation for operation as build_dobjs (rpt_api.hip) stores it in DObj.ox/oy/oz
inline void kernel_origin(const rpt_object &o, float p[3]) {
    const float vx = o.stationaryCam.y, vy = o.stationaryCam.z, vz = o.stationaryCam.w;
    for (int r = 0; r < 3; r++) p[r] = o.InvM[r].x * vx + o.InvM[r].y * vy + o.InvM[r].z * vz + o.InvM[r].w * 1.0f;
}

// Sets up A, ERR, the origin and B'; checks the global conditions of section 3.
inline bool setup(const rpt_object &ob, int interval, const float *root_bounds, Problem &p) {
    if (interval != 0 && interval != -1) { p.reason = R_NONFINITE; return false; }
    p.interval = interval;
    double I[3][3], L[3][4];
    for (int r = 0; r < 3; r++) {
        I[r][0] = ob.InvM[r].x; I[r][1] = ob.InvM[r].y; I[r][2] = ob.InvM[r].z;
        L[r][0] = ob.Lorentz[r + 1].x; L[r][1] = ob.Lorentz[r + 1].y; L[r][2] = ob.Lorentz[r + 1].z; L[r][3] = ob.Lorentz[r + 1].w;
    }
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) if (!(std::fabs(I[r][c]) < 1.0e15)) { p.reason = R_NONFINITE; return false; }      // (NaN fails too)
        for (int c = 0; c < 4; c++) if (!(std::fabs(L[r][c]) < 1.0e15)) { p.reason = R_NONFINITE; return false; }
    }
    float of[3];
    kernel_origin(ob, of);
    for (int k = 0; k < 3; k++) {
        p.o[k] = of[k];
        if (!(std::fabs(p.o[k]) < 1.0e15)) { p.reason = R_NONFINITE; return false; }
    }
    for (int i = 0; i < 3; i++) {
        double e = 0.0;
        for (int k = 0; k < 3; k++) {
            double row = 0.0;
            for (int j = 0; j < 4; j++) row += std::fabs(L[k][j]);
            e += std::fabs(I[i][k]) * row;
        }
        p.err[i] = 24.0 * U24 * e * (1.0 + 1.0e-12);
        for (int j = 0; j < 4; j++) p.A[i][j] = I[i][0] * L[0][j] + I[i][1] * L[1][j] + I[i][2] * L[2][j];
    }
    // the homeomorphism and noise conditions: A3 = A[.][1..3], c = A3^-1 A[.][0] (interval = -1), r0 = (1 - |c|) / ||A3^-1||_F
    double A3[3][3], A3i[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A3[i][j] = p.A[i][j + 1];
    if (!detail::invert3(A3, A3i)) { p.reason = R_NOT_HOMEOMORPHIC; return false; }
    double cn = 0.0;
    if (interval != 0) {
        double c[3];
        for (int i = 0; i < 3; i++) c[i] = A3i[i][0] * p.A[0][0] + A3i[i][1] * p.A[1][0] + A3i[i][2] * p.A[2][0];
        cn = std::sqrt(dot3(c, c));
    }
    // (the double evaluation of c is good to ~1e-15 * cond(A3); a margin of 1e-6 on |c| < 1 is far beyond that for every matrix
    // that passes the noise condition below)
    if (!(cn < 1.0 - 1.0e-6)) { p.reason = R_NOT_HOMEOMORPHIC; return false; }
    // the noise condition, in coordinates x_i / ERR_i (the argument of section 3 is invariant under a linear change of the object
    // coordinates; in these the error box is the unit cube, |e| <= sqrt 3, and an object scaled 100 : 2 : 0.01 is judged by its
    // boost, not by its shape):  r0 = (1 - |c|) / || A3^-1 diag(ERR) ||_F  must be at least 4 sqrt 3
    double fro = 0.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) fro += A3i[i][j] * p.err[j] * A3i[i][j] * p.err[j];
    fro = std::sqrt(fro);
    if (!std::isfinite(fro) || !(4.0 * 1.7320508075688773 * fro <= (1.0 - cn))) { p.reason = R_NOISE; return false; }
    // B' of section 2
    if (ob.type == RPT_SPHERE) {
        p.kind = 0;
        const double oo = dot3(p.o, p.o);
        p.R2 = 1.0 + 32.0 * U24 * (oo + 1.0);
        p.centre[0] = p.centre[1] = p.centre[2] = 0.0;
        if (!(oo >= 1.21 * p.R2)) { p.reason = R_ORIGIN_NEAR; return false; }        // |o| >= 1.1 R' (the forward-ray argument of section 2)
        return true;
    }
    p.kind = 1;
    if (ob.type == RPT_CUBE) {
        const double m = 1.0 + 4.0 * U24 * (1.0 + std::max(std::fabs(p.o[0]), std::max(std::fabs(p.o[1]), std::fabs(p.o[2]))));
        for (int k = 0; k < 3; k++) { p.lo[k] = -m; p.hi[k] = m; }
    } else if (ob.type == RPT_MESH && root_bounds) {
        for (int k = 0; k < 3; k++) {
            const double a = root_bounds[k], b = root_bounds[k + 3];
            if (!(std::fabs(a) < 1.0e15) || !(std::fabs(b) < 1.0e15) || !(a <= b)) { p.reason = R_NONFINITE; return false; }
            const double g = 4.0 * U24 * (std::max(std::fabs(a), std::fabs(b)) + std::fabs(p.o[k])) + 1.0e-30;
            p.lo[k] = a - g; p.hi[k] = b + g;
        }
    } else {
        p.reason = R_NONFINITE;
        return false;
    }
    // the origin must lie outside B' with a margin (a thousandth of the box's size): the cone K' is then pointed, the cube's
    // winding is +1 and the walk's "origin inside the root" branch (opencl_kernel.cl:233-248) is not taken
    bool outside = false;
    for (int k = 0; k < 3; k++) {
        p.centre[k] = 0.5 * (p.lo[k] + p.hi[k]);
        const double mg = 1.0e-3 * (p.hi[k] - p.lo[k]) + 1.0e-9 * (std::fabs(p.lo[k]) + std::fabs(p.hi[k])) + 1.0e-30;
        outside = outside || p.o[k] < p.lo[k] - mg || p.o[k] > p.hi[k] + mg;
    }
    if (!outside) { p.reason = R_ORIGIN_NEAR; return false; }
    return true;
}

// min over p in B' of n . (p - o), divided by nothing: > 0 means B' lies strictly on the positive side of the plane through o
inline double shape_side(const Problem &p, const double n[3]) {
    if (p.kind == 0) return -dot3(n, p.o) - std::sqrt(p.R2 * dot3(n, n));
    double s = 0.0;
    for (int k = 0; k < 3; k++) s += n[k] > 0.0 ? n[k] * (p.lo[k] - p.o[k]) : n[k] * (p.hi[k] - p.o[k]);
    return s;
}
// the point of B' - o that minimises n . x (a support point; for Gilbert's iteration)
inline void shape_support(const Problem &p, const double n[3], double x[3]) {
    if (p.kind == 0) {
        const double nn = std::sqrt(dot3(n, n)), R = std::sqrt(p.R2);
        for (int k = 0; k < 3; k++) x[k] = -p.o[k] - (nn > 0.0 ? R * n[k] / nn : 0.0);
        return;
    }
    for (int k = 0; k < 3; k++) x[k] = (n[k] > 0.0 ? p.lo[k] : p.hi[k]) - p.o[k];
}
// does the exact ray from o along g meet B'?  (used for the early "hit on the boundary" exit and, with a shrunken shape, the witness)
inline bool ray_meets(const Problem &p, const double g[3], double shrink) {
    if (p.kind == 0) {
        const double gg = dot3(g, g), b = -dot3(p.o, g);
        if (!(gg > 0.0) || !(b > 0.0)) return false;
        return dot3(p.o, p.o) - b * b / gg <= p.R2 * shrink * shrink;
    }
    double t0 = 0.0, t1 = 1.0e300;
    for (int k = 0; k < 3; k++) {
        const double h = 0.5 * (p.hi[k] - p.lo[k]) * shrink, lo = p.centre[k] - h, hi = p.centre[k] + h;
        if (g[k] == 0.0) {
            if (p.o[k] < lo || p.o[k] > hi) return false;
            continue;
        }
        double ta = (lo - p.o[k]) / g[k], tb = (hi - p.o[k]) / g[k];
        if (ta > tb) std::swap(ta, tb);
        t0 = std::max(t0, ta);
        t1 = std::min(t1, tb);
    }
    return t0 <= t1;
}

inline void generator(const Problem &p, double s, double fu, double fv, double g[3]) {
    for (int i = 0; i < 3; i++) g[i] = p.A[i][0] * s + p.A[i][1] * fu + p.A[i][2] * fv + p.A[i][3] * 0.5;
}

// ---- minimum-norm point of the convex hull of up to four points of R^3 (the sub-problem of the GJK iteration below; after
// C. Ericson's closest-point routines).  The simplex is reduced to the face that carries the closest point.
struct Simplex { double x[4][3]; int n = 0; };
inline void sub3(const double a[3], const double b[3], double r[3]) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
inline void cross3(const double a[3], const double b[3], double r[3]) { r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0]; }
inline void closest_on_segment(Simplex &s, double out[3]) {
    double ab[3];
    sub3(s.x[1], s.x[0], ab);
    const double den = dot3(ab, ab);
    double t = den > 0.0 ? -dot3(s.x[0], ab) / den : 0.0;
    if (t <= 0.0) { for (int i = 0; i < 3; i++) out[i] = s.x[0][i]; s.n = 1; return; }
    if (t >= 1.0) { for (int i = 0; i < 3; i++) { out[i] = s.x[1][i]; s.x[0][i] = s.x[1][i]; } s.n = 1; return; }
    for (int i = 0; i < 3; i++) out[i] = s.x[0][i] + t * ab[i];
}
// closest point of triangle (a, b, c) to the origin; `keep` receives which vertices carry it (bit mask)
inline void closest_on_triangle(const double a[3], const double b[3], const double c[3], double out[3], int &keep) {
    double ab[3], ac[3];
    sub3(b, a, ab);
    sub3(c, a, ac);
    const double d1 = -dot3(ab, a), d2 = -dot3(ac, a);
    if (d1 <= 0.0 && d2 <= 0.0) { for (int i = 0; i < 3; i++) out[i] = a[i]; keep = 1; return; }
    const double d3 = -dot3(ab, b), d4 = -dot3(ac, b);
    if (d3 >= 0.0 && d4 <= d3) { for (int i = 0; i < 3; i++) out[i] = b[i]; keep = 2; return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { const double v = d1 / (d1 - d3); for (int i = 0; i < 3; i++) out[i] = a[i] + v * ab[i]; keep = 3; return; }
    const double d5 = -dot3(ab, c), d6 = -dot3(ac, c);
    if (d6 >= 0.0 && d5 <= d6) { for (int i = 0; i < 3; i++) out[i] = c[i]; keep = 4; return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { const double w = d2 / (d2 - d6); for (int i = 0; i < 3; i++) out[i] = a[i] + w * ac[i]; keep = 5; return; }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int i = 0; i < 3; i++) out[i] = b[i] + w * (c[i] - b[i]);
        keep = 6;
        return;
    }
    const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
    for (int i = 0; i < 3; i++) out[i] = a[i] + ab[i] * v + ac[i] * w;
    keep = 7;
}
// false: the origin lies inside the tetrahedron (the two sets cannot be separated by a plane through the apex)
inline bool closest_on_simplex(Simplex &s, double out[3]) {
    if (s.n == 1) { for (int i = 0; i < 3; i++) out[i] = s.x[0][i]; return true; }
    if (s.n == 2) { closest_on_segment(s, out); return true; }
    auto reduce = [&](const int idx[3], int keep) {
        double y[3][3];
        int m = 0;
        for (int k = 0; k < 3; k++) if (keep & (1 << k)) { for (int i = 0; i < 3; i++) y[m][i] = s.x[idx[k]][i]; m++; }
        for (int k = 0; k < m; k++) for (int i = 0; i < 3; i++) s.x[k][i] = y[k][i];
        s.n = m;
    };
    if (s.n == 3) {
        int keep = 7;
        closest_on_triangle(s.x[0], s.x[1], s.x[2], out, keep);
        const int idx[3] = {0, 1, 2};
        reduce(idx, keep);
        return true;
    }
    // tetrahedron: the origin is inside iff, for every face, it lies on the side of the fourth vertex; otherwise the closest point
    // lies on the boundary, i.e. it is the best of the four faces' closest points (all four are tried: no visibility test to get wrong
    // on the needle-thin hulls a distant or flattened object produces)
    static const int faces[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
    double best = 1.0e300, bo[3] = {0, 0, 0};
    int bf = -1, bk = 7;
    bool inside = true;
    for (int f = 0; f < 4; f++) {
        const double *a = s.x[faces[f][0]], *b = s.x[faces[f][1]], *c = s.x[faces[f][2]], *d = s.x[faces[f][3]];
        double ab[3], ac[3], nrm[3], ad[3];
        sub3(b, a, ab);
        sub3(c, a, ac);
        cross3(ab, ac, nrm);
        sub3(d, a, ad);
        const double sd = dot3(ad, nrm), so = -dot3(a, nrm);
        inside = inside && sd * so > 0.0;
        double o3[3];
        int keep = 7;
        closest_on_triangle(a, b, c, o3, keep);
        const double q = dot3(o3, o3);
        if (q < best) { best = q; bf = f; bk = keep; for (int i = 0; i < 3; i++) bo[i] = o3[i]; }
    }
    if (inside) return false;
    if (bf < 0) return false;
    for (int i = 0; i < 3; i++) out[i] = bo[i];
    const int idx[3] = {faces[bf][0], faces[bf][1], faces[bf][2]};
    reduce(idx, bk);
    return true;
}

// THE segment test of section 3: true only if a normal n was found AND the two families of inequalities hold for it.
inline bool segment_separable(Problem &p, double ua, double va, double ub, double vb, bool &boundary_hit) {
    // exact range of |d| on the segment (|d|^2 = fu^2 + fv^2 + 1/4 is convex along it)
    const double du = ub - ua, dv = vb - va, dd = du * du + dv * dv;
    double tmin = dd > 0.0 ? -(ua * du + va * dv) / dd : 0.0;
    tmin = std::max(0.0, std::min(1.0, tmin));
    const double um = ua + tmin * du, vm = va + tmin * dv;
    const double dmin = std::sqrt(um * um + vm * vm + 0.25) * (1.0 - 1.0e-12);
    const double dmax = std::sqrt(std::max(ua * ua + va * va, ub * ub + vb * vb) + 0.25) * (1.0 + 1.0e-12);
    double G[4][3];
    int ng = 0;
    if (p.interval == 0) {
        generator(p, 0.0, ua, va, G[ng++]);
        generator(p, 0.0, ub, vb, G[ng++]);
    } else {
        const double s0 = p.interval * dmin, s1 = p.interval * dmax;
        generator(p, s0, ua, va, G[ng++]);
        generator(p, s0, ub, vb, G[ng++]);
        generator(p, s1, ua, va, G[ng++]);
        generator(p, s1, ub, vb, G[ng++]);
    }
    double E[3];
    for (int i = 0; i < 3; i++) E[i] = 2.0 * dmax * p.err[i];
    // scales that make the two point sets comparable for the minimum-norm iteration (they do not enter the check)
    double gs = 0.0;
    for (int k = 0; k < ng; k++) gs = std::max(gs, std::sqrt(dot3(G[k], G[k])));
    double cv[3] = {p.centre[0] - p.o[0], p.centre[1] - p.o[1], p.centre[2] - p.o[2]};
    const double cs = std::sqrt(dot3(cv, cv));
    if (!(gs > 0.0) || !(cs > 0.0) || !std::isfinite(gs) || !std::isfinite(cs)) return false;
    const double igs = 1.0 / gs, ics = 1.0 / cs;
    const double shape_scale = cs + std::sqrt(p.kind == 0 ? p.R2 : dot3(p.hi, p.hi) + dot3(p.lo, p.lo));
    // THE CHECK (this, and nothing else, is the proof): worst generator with the doubled error budget, and the shape's side
    double pad = 0.0, worst = 0.0, side = 0.0;
    int kw = 0;
    auto check = [&](const double n[3]) {
        const double nn = dot3(n, n);
        if (!(nn > 0.0) || !std::isfinite(nn)) return false;
        pad = 0.0;
        for (int i = 0; i < 3; i++) pad += std::fabs(n[i]) * E[i];
        worst = -1.0e300;
        for (int k = 0; k < ng; k++) {
            const double v = dot3(n, G[k]);
            if (v > worst) { worst = v; kw = k; }
        }
        side = shape_side(p, n);
        const double nl = std::sqrt(nn);
        return worst + pad < -1.0e-11 * nl * gs && side > 1.0e-11 * nl * shape_scale;
    };
    // Stage 1 — GJK in R^3 on conv( { -(G_k + e) / gs : |e_i| <= E_i } ∪ (B' - o) / cs ): its minimum-norm point n, if not 0, separates
    // the padded fan from the shape with the largest margin.  Every iterate is checked and accepted at once; good for everything
    // but needle-thin configurations (a fan and a shape that both subtend 1e-4 rad and pass 1e-6 rad from each other put the hull
    // within 1e-6 of the origin between points of norm 1: the iterate's DIRECTION is then lost to rounding).
    Simplex S;
    for (int i = 0; i < 3; i++) { S.x[0][i] = cv[i] * ics; S.x[1][i] = 0.0; }      // two points of the hull to start from: the shape's centre ...
    for (int k = 0; k < ng; k++) for (int i = 0; i < 3; i++) S.x[1][i] -= G[k][i] * igs / ng;      // ... and the fan's mean direction, reversed
    S.n = 2;
    double n[3];
    bool enclosed = false;
    for (int it = 0; it < GJK_STEPS; it++) {
        if (!closest_on_simplex(S, n)) { enclosed = true; break; }          // 0 inside the hull: not separable by a plane through o
        const double nn = dot3(n, n);
        if (!(nn > 1.0e-26) || !std::isfinite(nn)) break;
        if (check(n)) return true;
        double x[3];
        const double vg = -(worst + pad) * igs, vs = side * ics;
        if (vg < vs) {
            for (int i = 0; i < 3; i++) x[i] = -(G[kw][i] + (n[i] > 0.0 ? E[i] : -E[i])) * igs;
        } else {
            shape_support(p, n, x);
            for (int i = 0; i < 3; i++) x[i] *= ics;
        }
        if (std::min(vg, vs) >= nn * (1.0 - 1.0e-10)) break;      // no point of the hull is closer to 0
        bool known = false;                                       // the same support point again: rounding has taken over
        for (int k = 0; k < S.n; k++) known = known || (S.x[k][0] == x[0] && S.x[k][1] == x[1] && S.x[k][2] == x[2]);
        if (known || S.n >= 4) break;
        for (int i = 0; i < 3; i++) S.x[S.n][i] = x[i];
        S.n++;
    }
    {   // an exact ray through the segment's midpoint that meets B': nothing to certify here, and no halving will help
        const double mu = 0.5 * (ua + ub), mv = 0.5 * (va + vb);
        double g[3];
        generator(p, p.interval * std::sqrt(mu * mu + mv * mv + 0.25), mu, mv, g);
        if (ray_meets(p, g, 1.0)) { boundary_hit = true; return false; }
    }
    if (enclosed) return false;
    // Stage 2 — the same question on the plane: seen from o along the axis to the shape's centre, directions with a positive
    // component along the axis are points y = (x . e1, x . e2) / (x . axis) of a plane, planes through o are lines, the fan and the
    // shape are two convex sets of size ~ their angular size, and GJK on their Minkowski difference is well conditioned at any
    // scale.  A separating line m . y = tau gives the normal n = m1 e1 + m2 e2 - tau axis, which goes through the same check.
    {
        double ax[3] = {cv[0] * ics, cv[1] * ics, cv[2] * ics}, e1[3], e2[3];
        const double helper[3] = {std::fabs(ax[0]) < 0.6 ? 1.0 : 0.0, std::fabs(ax[0]) < 0.6 ? 0.0 : 1.0, 0.0};
        cross3(ax, helper, e1);
        const double l1 = std::sqrt(dot3(e1, e1));
        if (!(l1 > 0.0)) return false;
        for (int i = 0; i < 3; i++) e1[i] /= l1;
        cross3(ax, e1, e2);
        double fa[4][2], sb[8][2], rho = 0.0;
        int nb = 0;
        for (int k = 0; k < ng; k++) {
            const double z = dot3(G[k], ax);
            if (!(z > 1.0e-9 * gs)) return false;
            fa[k][0] = dot3(G[k], e1) / z; fa[k][1] = dot3(G[k], e2) / z;
        }
        if (p.kind == 0) {
            const double oo = dot3(p.o, p.o);
            rho = std::sqrt(p.R2 / (oo - p.R2));           // tan of the tangent cone's half angle (|o|^2 >= 1.21 R'^2, setup())
        } else {
            for (int c = 0; c < 8; c++) {
                const double x[3] = {((c & 1) ? p.hi[0] : p.lo[0]) - p.o[0], ((c & 2) ? p.hi[1] : p.lo[1]) - p.o[1], ((c & 4) ? p.hi[2] : p.lo[2]) - p.o[2]};
                const double z = dot3(x, ax);
                if (!(z > 1.0e-9 * cs)) return false;
                sb[nb][0] = dot3(x, e1) / z; sb[nb][1] = dot3(x, e2) / z; nb++;
            }
        }
        // support point of D = shape - fan that minimises v . x
        auto support = [&](const double v[2], double w[2], double &fan_max, double &shape_min) {
            int ka = 0;
            fan_max = -1.0e300;
            for (int k = 0; k < ng; k++) { const double t = v[0] * fa[k][0] + v[1] * fa[k][1]; if (t > fan_max) { fan_max = t; ka = k; } }
            double b[2];
            if (p.kind == 0) {
                const double vl = std::sqrt(v[0] * v[0] + v[1] * v[1]);
                b[0] = -rho * v[0] / vl; b[1] = -rho * v[1] / vl;
                shape_min = -rho * vl;
            } else {
                int kb = 0;
                shape_min = 1.0e300;
                for (int k = 0; k < nb; k++) { const double t = v[0] * sb[k][0] + v[1] * sb[k][1]; if (t < shape_min) { shape_min = t; kb = k; } }
                b[0] = sb[kb][0]; b[1] = sb[kb][1];
            }
            w[0] = b[0] - fa[ka][0]; w[1] = b[1] - fa[ka][1];
        };
        double T[3][2], v[2];
        int tn = 1;
        {   // start: shape centre (the plane's origin) minus the fan's first point
            T[0][0] = -fa[0][0]; T[0][1] = -fa[0][1];
        }
        for (int it = 0; it < GJK_STEPS; it++) {
            // closest point of the simplex T to the origin (2-D), reducing T
            if (tn == 1) { v[0] = T[0][0]; v[1] = T[0][1]; }
            else if (tn == 2) {
                const double ab[2] = {T[1][0] - T[0][0], T[1][1] - T[0][1]};
                const double den = ab[0] * ab[0] + ab[1] * ab[1];
                const double t = den > 0.0 ? -(T[0][0] * ab[0] + T[0][1] * ab[1]) / den : 0.0;
                if (t <= 0.0) { v[0] = T[0][0]; v[1] = T[0][1]; tn = 1; }
                else if (t >= 1.0) { v[0] = T[1][0]; v[1] = T[1][1]; T[0][0] = T[1][0]; T[0][1] = T[1][1]; tn = 1; }
                else { v[0] = T[0][0] + t * ab[0]; v[1] = T[0][1] + t * ab[1]; }
            } else {
                // triangle: inside -> not separable; else the best of its three edges
                auto orient = [&](int i, int j) { return T[i][0] * T[j][1] - T[i][1] * T[j][0]; };
                const double c0 = orient(0, 1), c1 = orient(1, 2), c2 = orient(2, 0);
                if ((c0 >= 0.0 && c1 >= 0.0 && c2 >= 0.0) || (c0 <= 0.0 && c1 <= 0.0 && c2 <= 0.0)) return false;
                double best = 1.0e300, bv[2] = {0, 0}, bt = 0.0;
                int bi = 0, bj = 1;
                const int ed[3][2] = {{0, 1}, {1, 2}, {2, 0}};
                for (int e = 0; e < 3; e++) {
                    const int i = ed[e][0], j = ed[e][1];
                    const double ab[2] = {T[j][0] - T[i][0], T[j][1] - T[i][1]};
                    const double den = ab[0] * ab[0] + ab[1] * ab[1];
                    double t = den > 0.0 ? -(T[i][0] * ab[0] + T[i][1] * ab[1]) / den : 0.0;
                    t = std::max(0.0, std::min(1.0, t));
                    const double q[2] = {T[i][0] + t * ab[0], T[i][1] + t * ab[1]};
                    const double d2 = q[0] * q[0] + q[1] * q[1];
                    if (d2 < best) { best = d2; bv[0] = q[0]; bv[1] = q[1]; bi = i; bj = j; bt = t; }
                }
                v[0] = bv[0]; v[1] = bv[1];
                const double P0[2] = {T[bi][0], T[bi][1]}, P1[2] = {T[bj][0], T[bj][1]};
                if (bt <= 0.0) { T[0][0] = P0[0]; T[0][1] = P0[1]; tn = 1; }
                else if (bt >= 1.0) { T[0][0] = P1[0]; T[0][1] = P1[1]; tn = 1; }
                else { T[0][0] = P0[0]; T[0][1] = P0[1]; T[1][0] = P1[0]; T[1][1] = P1[1]; tn = 2; }
            }
            const double vv = v[0] * v[0] + v[1] * v[1];
            if (!(vv > 0.0) || !std::isfinite(vv)) return false;
            double w[2], fan_max, shape_min;
            support(v, w, fan_max, shape_min);
            if (shape_min > fan_max) {      // v separates already: try the line halfway between the two sets
                const double tau = 0.5 * (fan_max + shape_min);
                double n3[3];
                for (int i = 0; i < 3; i++) n3[i] = v[0] * e1[i] + v[1] * e2[i] - tau * ax[i];
                if (check(n3)) return true;
            }
            if (v[0] * w[0] + v[1] * w[1] >= vv * (1.0 - 1.0e-12)) return false;      // converged (and the check did not pass)
            bool known = false;
            for (int k = 0; k < tn; k++) known = known || (T[k][0] == w[0] && T[k][1] == w[1]);
            if (known) return false;
            T[tn][0] = w[0]; T[tn][1] = w[1]; tn++;
        }
    }
    return false;
}

inline bool certify_segment(Problem &p, double ua, double va, double ub, double vb, int depth) {
    if (p.reason != CERTIFIED) return false;
    if (++p.tests > MAX_TESTS) { p.reason = R_BUDGET; return false; }
    p.max_depth = std::max(p.max_depth, depth);
    bool boundary_hit = false;
#ifdef RPT_CERT_TRACE
    std::fprintf(stderr, "  segment depth %d (%.9g, %.9g) - (%.9g, %.9g)\n", depth, ua, va, ub, vb);
#endif
    if (segment_separable(p, ua, va, ub, vb, boundary_hit)) return true;
    if (boundary_hit) { p.reason = R_HIT_ON_BOUNDARY; return false; }
    if (depth >= MAX_DEPTH) { p.reason = R_BUDGET; return false; }
    const double um = 0.5 * (ua + ub), vm = 0.5 * (va + vb);
    return certify_segment(p, ua, va, um, vm, depth + 1) && certify_segment(p, um, vm, ub, vb, depth + 1);
}

struct Poly { double u[24], v[24]; int n = 0; };

// keep the part of the polygon with a u + b v <= c
inline void clip(Poly &q, double a, double b, double c) {
    if (q.n == 0) return;
    Poly r;
    for (int k = 0; k < q.n; k++) {
        const int k2 = (k + 1) % q.n;
        const double f1 = a * q.u[k] + b * q.v[k] - c, f2 = a * q.u[k2] + b * q.v[k2] - c;
        if (f1 <= 0.0 && r.n < 24) { r.u[r.n] = q.u[k]; r.v[r.n] = q.v[k]; r.n++; }
        if ((f1 < 0.0 && f2 > 0.0) || (f1 > 0.0 && f2 < 0.0)) {
            const double t = f1 / (f1 - f2);
            if (r.n < 24) { r.u[r.n] = q.u[k] + t * (q.u[k2] - q.u[k]); r.v[r.n] = q.v[k] + t * (q.v[k2] - q.v[k]); r.n++; }
        }
    }
    q = r;
}

// Is the claim `r` proven for this object?  (r = empty_rect(): "no pixel of any frame hits it"; r = full_rect(): nothing claimed.)
// `p`: the problem as setup() left it (p.reason != CERTIFIED: setup failed, nothing can be proven).
inline bool certify_problem(Problem &p, const rpt_object &ob, int interval, const Rect &r, Stats *stats = nullptr) {
    auto done = [&](bool ok) {
        if (stats) { stats->reason = ok ? CERTIFIED : (p.reason == CERTIFIED ? R_BUDGET : p.reason); stats->tests = p.tests; stats->max_depth = p.max_depth; stats->segments = p.segments; }
        return ok;
    };
    if (p.reason != CERTIFIED) return done(false);
    // The claimed region inside the window, as a convex polygon, pulled IN by 1e-9 (what is certified is then a little more than
    // needed, and the polygon's own rounding — 1e-16 — cannot matter).  Sides at +-3e38 are no sides.
    Poly q;
    q.n = 4;
    q.u[0] = -WINDOW_U; q.v[0] = -WINDOW_V; q.u[1] = WINDOW_U; q.v[1] = -WINDOW_V; q.u[2] = WINDOW_U; q.v[2] = WINDOW_V; q.u[3] = -WINDOW_U; q.v[3] = WINDOW_V;
    const double IN = 1.0e-9, BIG = 1.0e30;
    const double hp[8][3] = {{-1, 0, -(double)r.u0}, {1, 0, (double)r.u1}, {0, -1, -(double)r.v0}, {0, 1, (double)r.v1},
                             {-1, -1, -(double)r.p_lo}, {1, 1, (double)r.p_hi}, {-1, 1, -(double)r.m_lo}, {1, -1, (double)r.m_hi}};
    for (int k = 0; k < 8; k++) {
        if (!(hp[k][2] == hp[k][2])) { p.reason = R_NONFINITE; return done(false); }
        if (hp[k][2] > BIG) continue;
        clip(q, hp[k][0], hp[k][1], hp[k][2] - IN * (k < 4 ? 1.0 : 1.5));
    }
    if (q.n < 3) q.n = 0;
    // the witness: the direction towards the shape's centre must be a hit of the shape's inner half and must not lie in P
    {
        detail::DirMap m = detail::make_map(ob, interval);
        detail::D3 nd;
        const detail::D3 uc{p.centre[0] - p.o[0], p.centre[1] - p.o[1], p.centre[2] - p.o[2]};
        if (!m.ok || !m.G(uc, nd)) { p.reason = R_WITNESS; return done(false); }
        const double l = std::sqrt(nd.x * nd.x + nd.y * nd.y + nd.z * nd.z);
        if (!(l > 0.0) || !std::isfinite(l)) { p.reason = R_WITNESS; return done(false); }
        const double w[3] = {nd.x / l, nd.y / l, nd.z / l};
        double g[3];
        for (int i = 0; i < 3; i++) g[i] = p.A[i][0] * interval + p.A[i][1] * w[0] + p.A[i][2] * w[1] + p.A[i][3] * w[2];
        if (!ray_meets(p, g, 0.5)) { p.reason = R_WITNESS; return done(false); }
        if (w[2] > 1.0e-9) {                                        // in front of the camera: where on the plane?
            const double fu = 0.5 * w[0] / w[2], fv = 0.5 * w[1] / w[2];
            const double MG = 1.0e-7;
            const bool clearly_outside_window = std::fabs(fu) > WINDOW_U + MG || std::fabs(fv) > WINDOW_V + MG;
            if (!clearly_outside_window) {
                bool inside_claim = q.n >= 3;
                for (int k = 0; k < 8 && inside_claim; k++) {
                    if (hp[k][2] > BIG) continue;
                    inside_claim = hp[k][0] * fu + hp[k][1] * fv <= hp[k][2] - IN * 2.0 - MG;
                }
                if (!inside_claim) { p.reason = R_WITNESS; return done(false); }
            }
        }
    }
    // the boundary of P: (i) the polygon's edges that do not lie on a side of the window, (ii) the window's sides outside the polygon
    auto on_side = [&](double u, double v, int side) {
        return side == 0 ? u == -WINDOW_U : side == 1 ? u == WINDOW_U : side == 2 ? v == -WINDOW_V : v == WINDOW_V;
    };
    for (int k = 0; k < q.n; k++) {
        const int k2 = (k + 1) % q.n;
        bool along = false;
        for (int s = 0; s < 4; s++) along = along || (on_side(q.u[k], q.v[k], s) && on_side(q.u[k2], q.v[k2], s));
        if (along) continue;
        p.segments++;
        if (!certify_segment(p, q.u[k], q.v[k], q.u[k2], q.v[k2], 0)) return done(false);
    }
    for (int s = 0; s < 4; s++) {
        const bool vertical = s < 2;                                 // a side u = const runs along v, a side v = const along u
        const double fixed = s == 0 ? -WINDOW_U : s == 1 ? WINDOW_U : s == 2 ? -WINDOW_V : WINDOW_V, half = vertical ? WINDOW_V : WINDOW_U;
        double lo = 1.0e300, hi = -1.0e300;
        for (int k = 0; k < q.n; k++)
            if (on_side(q.u[k], q.v[k], s)) { const double x = vertical ? q.v[k] : q.u[k]; lo = std::min(lo, x); hi = std::max(hi, x); }
        double pieces[2][2];
        int np = 0;
        if (lo > hi) { pieces[np][0] = -half; pieces[np][1] = half; np++; }
        else {
            if (lo > -half) { pieces[np][0] = -half; pieces[np][1] = lo; np++; }
            if (hi < half) { pieces[np][0] = hi; pieces[np][1] = half; np++; }
        }
        for (int k = 0; k < np; k++) {
            p.segments++;
            const bool ok = vertical ? certify_segment(p, fixed, pieces[k][0], fixed, pieces[k][1], 0)
                                     : certify_segment(p, pieces[k][0], fixed, pieces[k][1], fixed, 0);
            if (!ok) return done(false);
        }
    }
    return done(true);
}

inline bool certify(const rpt_object &ob, int interval, const float *root_bounds, const Rect &r, Stats *stats = nullptr) {
    Problem p;
    setup(ob, interval, root_bounds, p);
    return certify_problem(p, ob, interval, r, stats);
}

}  // namespace cert

// The proposal for object `o`: the outline sampling of rpt_screen_bounds.hpp, run on the shape B' the proof is about (the cube or
// the root box grown by the intersector's float error, rpt_bounds_certify.hpp section 2 — a proposal drawn around a smaller shape
// could not be proven where that growth shows on the screen: a cube seen from 8 000 of its own sizes away).
inline Rect proposed_object_rect(const rpt_object &o, int interval, const float *root_bounds, cert::Problem *problem_out = nullptr) {
    cert::Problem p;
    const bool ok = cert::setup(o, interval, root_bounds, p);
    if (problem_out) *problem_out = p;
    if (!ok) return full_rect();                       // (origin inside or near the shape, non-finite input ...: nothing could be proven)
    if (p.kind == 0) return sphere_rect(o, interval);
    return box_rect(o, interval, p.lo, p.hi);
}

// The region the kernel may use for object `o`: the proposal if it can be PROVEN, else the full plane.
inline Rect certified_object_rect(const rpt_object &o, int interval, const float *root_bounds, cert::Stats *stats = nullptr) {
    cert::Problem p;
    const Rect r = proposed_object_rect(o, interval, root_bounds, &p);
    if (r.u0 <= -3.0e38f && r.v0 <= -3.0e38f && r.u1 >= 3.0e38f && r.v1 >= 3.0e38f && !has_diagonals(r)) {
        if (stats) *stats = cert::Stats{-1, 0, 0, 0};      // nothing was claimed
        return r;
    }
    return cert::certify_problem(p, o, interval, r, stats) ? r : full_rect();
}

}  // namespace rptb
