// rpt_screen_bounds.hpp — per object, per frame: a conservative rectangle on the camera's image plane outside
// which no primary ray can reach the object.  Host code (double precision), run in rpt_set_objects; the render
// kernel turns the rectangles into a per-wavefront object mask with ONE lane-parallel overlap test and a __ballot
// (rpt_kernels.hip.h), so no pixel-space prepass and no second launch is needed.
//
// What is bounded is the reference's own ray set-up (opencl_kernel.cl:382-390): a primary ray with camera direction
// nd is, for object i, the object-space ray   origin oc = InvM * stationaryCam.yzw,
//                                             dir    F(nd) = InvM3 * (Lorentz * (interval, nd))[1..3].
// The object lies inside a bounding shape B in object space (the unit sphere, the cube [-1,1]^3, a mesh's root box), so a
// ray can only hit it if F(nd) lies in K = { directions from oc that meet B }, a convex cone.  F is a homeomorphism of the
// direction sphere (aberration = a Moebius map, then a linear map), so the region of camera directions to be kept is
// bounded by G(boundary of K), G = F^-1:  the silhouette curve of B is sampled (adaptively, where the IMAGE of the curve
// needs the samples: struct Curve below), every sample is mapped to a camera direction, a subset is CHECKED by pushing
// it through F again with the very matrices the kernel uses, and the rectangle is the
// bounding box of the samples' image-plane positions (plane z = 0.5: u = x/2z, v = y/2z), grown by a margin that covers
// the curve between samples.  The region is cut off at the cone nd.z = EPS |nd| ("the horizon": a circle of radius ~25 on
// the plane, far outside any screen): where the outline crosses it the crossing point is located, and the part of the
// horizon circle that lies INSIDE the region — decided for sampled horizon directions by sending them through F and
// asking whether that object-space ray meets B — is added to the box, so a floor under the camera or an object half
// behind it gets a rectangle that is open exactly on the sides where it runs off to infinity.  What is boxed is then the
// complete boundary of a bounded planar region, so no inside/outside question is left open.  Every doubtful case — origin
// inside or near B, non-finite numbers, a failed check, an outline the sampling cannot resolve, an object so far away or so
// strongly boosted that the kernel's own float arithmetic no longer follows the geometry — returns the full plane.
// Arithmetic here may be approximate: it only decides which exact tests are skipped, and a skipped test is one the
// reference would have failed for every pixel.  (What "would have failed" means is decided by the kernel's FLOAT results,
// not by exact geometry: see sphere_rect.)
#pragma once
#include <algorithm>
#include <cmath>

#include "../../include/rpt_layout.h"

namespace rptb {

// Keep the object for a tile iff the tile's plane rectangle overlaps [u0,u1] x [v0,v1] AND reaches into the two diagonal
// slabs p_lo <= u + v <= p_hi, m_lo <= u - v <= m_hi (an octagon: the rectangle with its corners cut where that pays —
// a floor's horizon bent into a V by aberration, a ruler lying diagonally).  The diagonal bounds hold for the part of the
// region inside the window |u| <= DIAG_WINDOW_U, |v| <= DIAG_WINDOW_V only (they are taken from the outline clipped to it,
// so that what an outline does on its way to the horizon cannot spoil them): the kernel uses them for frames that fit it.
struct Rect { float u0, v0, u1, v1, p_lo, p_hi, m_lo, m_hi; };
constexpr double DIAG_WINDOW_U = 2.0, DIAG_WINDOW_V = 0.55;      // frames up to 4 : 1

inline Rect full_rect() { return Rect{-3.0e38f, -3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f}; }
inline Rect empty_rect() { return Rect{3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f}; }
inline bool has_diagonals(const Rect &r) { return r.p_lo > -3.0e38f || r.p_hi < 3.0e38f || r.m_lo > -3.0e38f || r.m_hi < 3.0e38f; }

namespace detail {

constexpr double EPS_FRONT = 0.02;          // directions with nd.z <= EPS |nd| are treated as behind the camera

struct D3 { double x, y, z; };
inline D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 mul(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double len(D3 a) { return std::sqrt(dot(a, a)); }
inline bool finite3(D3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }

inline bool invert3(const double m[3][3], double out[3][3]) {
    const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2], c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    const double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
    if (!std::isfinite(det) || std::fabs(det) < 1.0e-300) return false;
    const double id = 1.0 / det;
    out[0][0] = c00 * id; out[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id; out[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    out[1][0] = c01 * id; out[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id; out[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    out[2][0] = c02 * id; out[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id; out[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
    return true;
}

// The two maps of one object.
struct DirMap {
    double L[4][4], Linv[4][4], LsInv[3][3], M3[3][3], InvM3[3][3];
    int interval = -1;
    bool ok = false;
    bool linear = false;      // G(u) (before normalisation) is linear in u: straight object-space edges stay straight on the image plane

    // camera direction (any length) -> object-space direction, with the kernel's matrices (opencl_kernel.cl:386-388, 214)
    D3 F(D3 nd) const {
        const double l = len(nd);
        const double v[4] = {(double)interval, nd.x / l, nd.y / l, nd.z / l};
        double r[3];
        for (int i = 0; i < 3; i++) r[i] = L[i + 1][0] * v[0] + L[i + 1][1] * v[1] + L[i + 1][2] * v[2] + L[i + 1][3] * v[3];
        return {InvM3[0][0] * r[0] + InvM3[0][1] * r[1] + InvM3[0][2] * r[2], InvM3[1][0] * r[0] + InvM3[1][1] * r[1] + InvM3[1][2] * r[2],
                InvM3[2][0] * r[0] + InvM3[2][1] * r[1] + InvM3[2][2] * r[2]};
    }
    // object-space direction -> camera direction (not normalised); false when no such direction exists
    bool G(D3 u, D3 &nd) const {
        const double r[3] = {M3[0][0] * u.x + M3[0][1] * u.y + M3[0][2] * u.z, M3[1][0] * u.x + M3[1][1] * u.y + M3[1][2] * u.z,
                             M3[2][0] * u.x + M3[2][1] * u.y + M3[2][2] * u.z};
        if (interval == 0) {
            nd = {LsInv[0][0] * r[0] + LsInv[0][1] * r[1] + LsInv[0][2] * r[2], LsInv[1][0] * r[0] + LsInv[1][1] * r[1] + LsInv[1][2] * r[2],
                  LsInv[2][0] * r[0] + LsInv[2][1] * r[1] + LsInv[2][2] * r[2]};
            return finite3(nd) && dot(nd, nd) > 0.0;
        }
        // interval = -1: (interval, nd) is a past-directed null vector and stays one under Lorentz
        const double rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const double k[4] = {-rl, r[0], r[1], r[2]};
        double c[4];
        for (int i = 0; i < 4; i++) c[i] = Linv[i][0] * k[0] + Linv[i][1] * k[1] + Linv[i][2] * k[2] + Linv[i][3] * k[3];
        nd = {c[1], c[2], c[3]};
        return c[0] < 0.0 && finite3(nd) && dot(nd, nd) > 0.0;
    }
    D3 to_rest(D3 u) const {      // object-space direction -> direction in the object's rest frame
        return {M3[0][0] * u.x + M3[0][1] * u.y + M3[0][2] * u.z, M3[1][0] * u.x + M3[1][1] * u.y + M3[1][2] * u.z,
                M3[2][0] * u.x + M3[2][1] * u.y + M3[2][2] * u.z};
    }
    // G, verified through F: the angle between F(G(u)) and u must vanish — against `tol`, the sine of an angle that is
    // small next to the bounding shape's angular size as seen from the camera (an absolute tolerance says nothing when a
    // strongly boosted object subtends 1e-3 rad in its own frame and half the screen in the camera's)
    double tol = 1.0e-4;
    bool verify(D3 u, D3 nd) const {
        const D3 back = F(nd);
        const double lb = len(back), lu = len(u);
        if (!(lb > 0.0) || !(lu > 0.0) || !std::isfinite(lb)) return false;
        return dot(back, u) > 0.0 && len(cross(back, u)) <= tol * lb * lu;
    }
    bool G_checked(D3 u, D3 &nd) const { return G(u, nd) && verify(u, nd); }
    // The kernel evaluates F in float: the boosted null vector (-1, nd) has time and space parts of size gamma that cancel
    // down to gamma (1 - beta cos) >= 1 / (2 gamma), so its direction carries a relative error of up to ~2 gamma^2 ulp.
    // Is that small (half a percent) next to an angular size `ang` (radians, object space)?  If not, no statement is made.
    bool float_noise_small_against(double ang) const {
        if (interval == 0) return true;
        const double g = L[0][0];
        return std::isfinite(g) && 2.0e-7 * g * g <= 0.005 * ang;
    }
};

inline DirMap make_map(const rpt_object &o, int interval) {
    DirMap m;
    m.interval = interval;
    if (interval != 0 && interval != -1) return m;
    const rpt_float4 *Lr = o.Lorentz, *Mr = o.M, *Ir = o.InvM;
    for (int r = 0; r < 4; r++) { m.L[r][0] = Lr[r].x; m.L[r][1] = Lr[r].y; m.L[r][2] = Lr[r].z; m.L[r][3] = Lr[r].w; }
    for (int r = 0; r < 3; r++) {
        m.M3[r][0] = Mr[r].x; m.M3[r][1] = Mr[r].y; m.M3[r][2] = Mr[r].z;
        m.InvM3[r][0] = Ir[r].x; m.InvM3[r][1] = Ir[r].y; m.InvM3[r][2] = Ir[r].z;
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) if (!std::isfinite(m.L[r][c])) return m;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) if (!std::isfinite(m.M3[r][c]) || !std::isfinite(m.InvM3[r][c])) return m;
    if (interval == 0) {
        double ls[3][3];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) ls[r][c] = m.L[r + 1][c + 1];
        if (!invert3(ls, m.LsInv)) return m;
        m.linear = true;
    } else {
        // the caller's InvLorentz serves as the inverse (the kernel shades with it): every use below is checked through F
        const rpt_float4 *Jr = o.InvLorentz;
        for (int r = 0; r < 4; r++) { m.Linv[r][0] = Jr[r].x; m.Linv[r][1] = Jr[r].y; m.Linv[r][2] = Jr[r].z; m.Linv[r][3] = Jr[r].w; }
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) if (!std::isfinite(m.Linv[r][c])) return m;
        m.linear = m.Linv[1][0] == 0.0 && m.Linv[2][0] == 0.0 && m.Linv[3][0] == 0.0;
    }
    m.ok = true;
    return m;
}

// Accumulates the image-plane bounding box of a sampled closed or open curve of object-space directions.
struct Accum {
    const DirMap &m;
    Accum &operator=(const Accum &o) {
        u0 = o.u0; v0 = o.v0; u1 = o.u1; v1 = o.v1; failed = o.failed; any_front = o.any_front; any_near_behind = o.any_near_behind;
        n_crossings = o.n_crossings; p_lo = o.p_lo; p_hi = o.p_hi; m_lo = o.m_lo; m_hi = o.m_hi; pen = o.pen; pen_u = o.pen_u; pen_v = o.pen_v;
        return *this;
    }
    double u0 = 1e300, v0 = 1e300, u1 = -1e300, v1 = -1e300;
    // extents of u + v and u - v over the outline clipped to the diagonal window; the outline is fed as polylines
    double p_lo = 1e300, p_hi = -1e300, m_lo = 1e300, m_hi = -1e300;
    bool pen = false;
    double pen_u = 0.0, pen_v = 0.0;
    void diag_point(double u, double v) {
        p_lo = std::min(p_lo, u + v); p_hi = std::max(p_hi, u + v); m_lo = std::min(m_lo, u - v); m_hi = std::max(m_hi, u - v);
    }
    static bool in_window(double u, double v) { return u >= -DIAG_WINDOW_U && u <= DIAG_WINDOW_U && v >= -DIAG_WINDOW_V && v <= DIAG_WINDOW_V; }
    void diag_segment(double ua, double va, double ub, double vb) {      // Liang-Barsky clip to the window, then both end points
        if (in_window(ua, va) && in_window(ub, vb)) {       // the usual case: nothing to clip (the window is convex)
            diag_point(ua, va);
            diag_point(ub, vb);
            return;
        }
        double t0 = 0.0, t1 = 1.0;
        const double du = ub - ua, dv = vb - va;
        const double pp[4] = {-du, du, -dv, dv}, qq[4] = {ua + DIAG_WINDOW_U, DIAG_WINDOW_U - ua, va + DIAG_WINDOW_V, DIAG_WINDOW_V - va};
        for (int k = 0; k < 4; k++) {
            if (pp[k] == 0.0) { if (qq[k] < 0.0) return; continue; }
            const double r = qq[k] / pp[k];
            if (pp[k] < 0.0) { if (r > t1) return; t0 = std::max(t0, r); }
            else { if (r < t0) return; t1 = std::min(t1, r); }
        }
        if (!(t0 <= t1)) return;
        diag_point(ua + du * t0, va + dv * t0);
        diag_point(ua + du * t1, va + dv * t1);
    }
    void pen_up() { pen = false; }
    // the next point of the current polyline (a direction in front of the camera): box, and the segment from the last one
    void point(D3 nd) {
        const double iz = 0.5 / nd.z, u = nd.x * iz, v = nd.y * iz;
        u0 = std::min(u0, u); u1 = std::max(u1, u); v0 = std::min(v0, v); v1 = std::max(v1, v);
        if (pen) diag_segment(pen_u, pen_v, u, v);
        else diag_segment(u, v, u, v);
        pen = true; pen_u = u; pen_v = v;
        any_front = true;
    }
    bool failed = false, any_front = false, any_near_behind = false;
    int n_crossings = 0;
    explicit Accum(const DirMap &map) : m(map) {}

    // in front of the clip cone: nd.z > EPS |nd|, without the square root
    static bool is_front(D3 nd) { return nd.z > 0.0 && nd.z * nd.z > EPS_FRONT * EPS_FRONT * dot(nd, nd); }
    void take(D3 nd) {      // a lone direction in front of the camera (horizon samples: far outside the diagonal window)
        const double u = 0.5 * nd.x / nd.z, v = 0.5 * nd.y / nd.z;
        u0 = std::min(u0, u); u1 = std::max(u1, u); v0 = std::min(v0, v); v1 = std::max(v1, v);
    }
    // Map one sample of the outline: its camera direction in nd and whether that is in front of the camera; nothing is
    // recorded yet (the caller feeds crossings and points in curve order).  `verify`: push it through F again (done at
    // the end points of every edge and at every eighth sample of a rim; a wrong inverse shows there too).
    bool map_sample(D3 u_obj, D3 &nd, bool &front, bool verify) {
        if (!(verify ? m.G_checked(u_obj, nd) : m.G(u_obj, nd))) { failed = true; return false; }
        classify(nd, front);
        return true;
    }
    void classify(D3 nd, bool &front) {
        front = is_front(nd);
        if (!front && (nd.z >= 0.0 || nd.z * nd.z < 0.04 * dot(nd, nd))) any_near_behind = true;      // nd.z > -0.2 |nd|
    }
    // one more sample of the polyline being drawn, after map_sample/classify (and crossing(), if the side changed)
    void feed(D3 nd, bool front) {
        if (front) point(nd);
        else pen_up();
    }
    // the curve between two consecutive samples crosses the clip cone: locate the crossing by bisection on the
    // object-space segment between them (exact for straight edges, close enough for arcs: the margin covers it)
    void crossing(D3 ua, D3 ub, bool front_a) {
        D3 lo = ua, hi = ub;      // lo keeps ua's side, hi ub's
        for (int it = 0; it < 12; it++) {
            const D3 mid = mul(add(lo, hi), 0.5);
            D3 nd;
            if (!m.G(mid, nd)) { failed = true; return; }
            const bool f = is_front(nd);
            if (f == front_a) lo = mid; else hi = mid;
        }
        const D3 cross_u = front_a ? lo : hi, from_u = front_a ? ua : ub;      // the front side of the crossing; the front sample
        D3 xnd;
        if (!m.G_checked(cross_u, xnd) || !(xnd.z > 0.0)) { failed = true; return; }
        n_crossings++;
        // A curved outline runs off towards the horizon along an asymptote and may swing past both of its end points on the
        // way: it is followed between the sample in front and the crossing in steps that halve the distance to the crossing.
        D3 way[10];
        int n_way = 0;
        if (!m.linear) {
            double w = 0.5;
            for (int k = 0; k < 10; k++, w *= 0.5) {
                const D3 u = add(mul(cross_u, 1.0 - w), mul(from_u, w));
                D3 n2;
                if (!m.G(u, n2)) { failed = true; return; }
                if (is_front(n2)) way[n_way++] = n2;
            }
        }
        if (front_a) {              // the pen stands on the front sample: ... -> way[0] -> ... -> crossing, and up
            for (int k = 0; k < n_way; k++) point(way[k]);
            point(xnd);
            pen_up();
        } else {                    // a new stretch starts at the crossing and runs towards the front sample the caller adds next
            pen_up();
            point(xnd);
            for (int k = n_way - 1; k >= 0; k--) point(way[k]);
        }
    }
    // The horizon circle nd.z = EPS |nd|: which part of it lies inside the kept region?  `inside(d)` answers for an
    // object-space direction d (does the ray from oc along d meet the inflated bounding shape).  n directions are tried;
    // a member also claims its two neighbours, which covers the arc between samples.  Returns the number of members.
    template <class Inside>
    int horizon(int n, Inside inside) {
        bool member[64];
        if (n > 64) n = 64;
        const double s = std::sqrt(1.0 - EPS_FRONT * EPS_FRONT);
        int count = 0;
        for (int j = 0; j < n; j++) {
            const double phi = 2.0 * M_PI * j / n;
            member[j] = inside(m.F(D3{s * std::cos(phi), s * std::sin(phi), EPS_FRONT}));
            count += member[j];
        }
        for (int j = 0; j < n && count; j++) {
            if (!(member[j] || member[(j + 1) % n] || member[(j + n - 1) % n])) continue;
            const double phi = 2.0 * M_PI * j / n;
            take(D3{s * std::cos(phi), s * std::sin(phi), EPS_FRONT});
            any_front = true;
        }
        return count;
    }
    // An outline that lies wholly in front of the clip cone is a closed curve inside the front cap; the kept region is one of
    // the two parts it cuts the direction sphere into — the one inside the cap, or (under strong aberration) the one that
    // holds everything else, the horizon and the whole back cap included.  ONE direction of the back cap tells which.
    template <class Inside>
    bool wraps_behind(Inside inside) const { return inside(m.F(D3{0.0, 0.0, -1.0})); }
    // The diagonal extents are those of (region ∩ window): besides the outline clipped to the window, that set's boundary
    // can run along the window's own edges, where u + v and u - v are monotone — so it ends in a window corner (asked here)
    // or where the outline crosses the edge (already taken with the clipped outline).
    // (a region whose box lies strictly inside the window contains none of its corners)
    bool reaches_window_edge() const { return !(u0 > -DIAG_WINDOW_U && u1 < DIAG_WINDOW_U && v0 > -DIAG_WINDOW_V && v1 < DIAG_WINDOW_V); }
    template <class Inside>
    void window_corners(Inside inside) {
        for (int k = 0; k < 4; k++) {
            const double u = (k & 1) ? DIAG_WINDOW_U : -DIAG_WINDOW_U, v = (k & 2) ? DIAG_WINDOW_V : -DIAG_WINDOW_V;
            if (inside(m.F(D3{2.0 * u, 2.0 * v, 1.0}))) diag_point(u, v);
        }
    }
};

// One outline curve (a box edge, a sphere's rim), sampled ADAPTIVELY.  The curve is given by a parameter t -> object-space
// direction; how the camera sees it is anything but uniform in t — an edge that passes close to the camera, or any curve
// under aberration, can put half the screen between two neighbouring samples of a uniform grid (found by the extreme-scene
// soak: a 10 x 2.5 x 12 cube at 0.9c next to a camera at 0.5c, eight uniform segments per edge, hits 0.17 outside the
// diagonal bounds).  So the samples are placed where the IMAGE needs them: an interval is halved while its midpoint lies
// off the chord by more than 3 `tol` (tol = a quarter of the margin the bounds get; once the midpoint is in, a quarter of that
// deviation is what remains), while its two halves differ in length by more
// than 3 : 1 (the parameter runs unevenly: the far half hides a bulge), while a chord is longer than a quarter of the screen
// height, or — down to 1/16 of a base interval — while any of its three samples lies behind the clip cone (crossing() then
// bisects where the side changes); off the screen all of this is relative to the stretch's distance from the screen.  An interval that still wants halving at 1/512 of a base interval, or a curve of more than 640 samples,
// makes the whole object's bounds the full plane.  The samples come out in curve order for Accum's polyline logic.
struct Curve {
    struct Sample { D3 u, nd; bool front; double pu, pv; };
    static constexpr int CAP = 640, MAX_DEPTH = 9;
    Sample s[CAP];
    int n = 0;
    bool failed = false, clipped = false;
    double tol = 1.0e-3;

    template <class Param>
    Sample eval(Accum &acc, Param &at, double t, const D3 *known_u, const D3 *known_nd) {
        Sample r;
        r.u = known_u ? *known_u : at(t);
        r.front = false;
        r.pu = r.pv = 0.0;
        if (known_nd) { r.nd = *known_nd; acc.classify(r.nd, r.front); }
        else if (!acc.map_sample(r.u, r.nd, r.front, false)) { failed = true; r.nd = D3{0, 0, 1}; }
        if (r.front) { const double iz = 0.5 / r.nd.z; r.pu = r.nd.x * iz; r.pv = r.nd.y * iz; }
        clipped = clipped || !r.front;
        return r;
    }
    void push(const Sample &x) {
        if (n < CAP) s[n++] = x; else failed = true;
    }
    bool needs_more(const Sample &a, const Sample &m, const Sample &b, int depth) const {
        // a stretch that is not wholly in front of the clip cone is looked at in sixteenths of the base interval: where it comes
        // to the front (crossing() then bisects), and whether a stretch between two samples behind the camera does at all
        // ... and further wherever two neighbouring samples IN FRONT are still far apart on the image (the visible end of a stretch
        // that leaves through the clip cone inside one sixteenth)
        if (!(a.front && m.front && b.front)) {
            if (depth < 4) return true;
            auto off = [](const Sample &x) {
                const double dx = std::max(0.0, std::fabs(x.pu) - DIAG_WINDOW_U), dy = std::max(0.0, std::fabs(x.pv) - DIAG_WINDOW_V);
                return std::sqrt(dx * dx + dy * dy);
            };
            auto far_apart = [&](const Sample &x, const Sample &y) {
                if (!(x.front && y.front)) return false;
                const double lim = std::max(0.25, 0.5 * std::min(off(x), off(y))), dx = x.pu - y.pu, dy = x.pv - y.pv;
                return dx * dx + dy * dy > lim * lim;
            };
            // ... and a sample in front next to one behind must itself be well outside the window: between the two the curve runs
            // off to the clip cone, and the part of that run that crosses the window has to lie between samples that see it
            auto leaves_unwatched = [&](const Sample &x, const Sample &y) { return x.front != y.front && off(x.front ? x : y) < 3.0; };
            return far_apart(a, m) || far_apart(m, b) || leaves_unwatched(a, m) || leaves_unwatched(m, b);
        }
        // Off the screen the tolerance grows with the distance D of the three samples from the box |u| <= 2, |v| <= 0.55 (the window
        // the diagonal bounds are taken in — a slab that touches the region out at u = 1.8 still cuts pixels at u = 0.5):
        // what the bounds need from a stretch out there is only that it does not come in unnoticed, and a curve that runs off
        // towards the horizon (image coordinates up to 25) cannot be followed to a fraction of a pixel all the way.  (A first
        // version simply skipped stretches whose three samples lay beyond three screen heights on the same side — and lost an
        // edge that swung from u = -3.8 in to u = -0.23 and back between two of them, under a relative gamma of 17.)
        auto off_screen = [](const Sample &x) {
            const double dx = std::max(0.0, std::fabs(x.pu) - DIAG_WINDOW_U), dy = std::max(0.0, std::fabs(x.pv) - DIAG_WINDOW_V);
            return std::sqrt(dx * dx + dy * dy);
        };
        const double D = std::min(off_screen(a), std::min(off_screen(m), off_screen(b)));
        // (squared lengths throughout.)  With the midpoint inserted, what is left of a smooth arc's deviation is about a quarter
        // of what the midpoint showed: the interval is good once that is within tol, i.e. dev <= 3 tol ...
        const double cx = b.pu - a.pu, cy = b.pv - a.pv, cl2 = cx * cx + cy * cy;
        const double ax = m.pu - a.pu, ay = m.pv - a.pv, bx = b.pu - m.pu, by = b.pv - m.pv;
        const double la2 = ax * ax + ay * ay, lb2 = bx * bx + by * by;
        const double area = cx * ay - cy * ax;                          // |area| / |chord| = the midpoint's distance from the chord
        const double lmax2 = std::max(la2, lb2), lmin2 = std::min(la2, lb2);
        // ... for a GENTLE arc, that is (halves within 1.5 : 1, midpoint within 8 % of the chord's length off it: curvature about
        // even); anything else is held to tol itself, i.e. its halves get looked at
        const bool gentle = lmax2 <= 2.25 * lmin2 && area * area <= 0.0064 * cl2 * cl2;
        const double t3 = std::max(gentle ? 3.0 * tol : tol, 0.1 * D);
        if (!(cl2 > 0.0 ? area * area <= t3 * t3 * cl2 : la2 <= t3 * t3)) return true;      // (NaN refines too)
        if (lmax2 > 9.0 * lmin2 + 4.0 * t3 * t3) return true;           // halves more uneven than 3 : 1
        const double lmax = std::max(0.25, 0.5 * D);                     // a chord longer than a quarter of the screen height, or than half its distance
        return lmax2 > lmax * lmax;
    }
    template <class Param>
    void refine(Accum &acc, Param &at, const Sample &a, double ta, const Sample &b, double tb, int depth) {
        if (failed) return;
        const double tm = 0.5 * (ta + tb);
        const Sample m = eval(acc, at, tm, nullptr, nullptr);
        const bool wants = !failed && needs_more(a, m, b, depth);
        if (wants && depth >= MAX_DEPTH) failed = true;      // still not resolved at 1/512 of a base interval: no statement (full plane)
        const bool more = wants && !failed;
        if (more) refine(acc, at, a, ta, m, tm, depth + 1);
        push(m);
        if (more) refine(acc, at, m, tm, b, tb, depth + 1);
    }
    // Samples the curve over `intervals` (<= 16) base intervals of t in [0, 1]; first/last: known end points (or null).  The
    // tolerance is a quarter of the margin finish() will add: rel_margin of the on-screen extent — `extent` if the caller knows
    // it (a box: of all its corners), else that of the base samples — plus 1e-3.
    template <class Param>
    void sample(Accum &acc, Param at, int intervals, bool adaptive, const D3 *u0, const D3 *nd0, const D3 *u1, const D3 *nd1,
                double rel_margin, double extent, double forced_tol = -1.0) {
        n = 0;
        failed = false;
        clipped = false;
        Sample base[17];
        if (intervals > 16) intervals = 16;
        for (int k = 0; k <= intervals && !failed; k++)
            base[k] = k == 0 ? eval(acc, at, 0.0, u0, nd0) : (k == intervals ? eval(acc, at, 1.0, u1, nd1) : eval(acc, at, (double)k / intervals, nullptr, nullptr));
        if (!failed && adaptive) {
            if (!(extent > 0.0)) {
                double a0 = 1e300, a1 = -1e300, b0 = 1e300, b1 = -1e300;
                for (int k = 0; k <= intervals; k++) {
                    if (!base[k].front) continue;
                    const double pu = std::max(-1.25, std::min(1.25, base[k].pu)), pv = std::max(-0.55, std::min(0.55, base[k].pv));
                    a0 = std::min(a0, pu); a1 = std::max(a1, pu); b0 = std::min(b0, pv); b1 = std::max(b1, pv);
                }
                extent = a1 >= a0 ? std::min(a1 - a0, b1 - b0) : 0.0;        // the smaller one: each side's margin is relative to its own extent
            }
            tol = forced_tol > 0.0 ? forced_tol : std::max(2.5e-4, 0.25 * (rel_margin * extent + 1.0e-3));
        }
        for (int k = 0; k <= intervals && !failed; k++) {
            if (k > 0 && adaptive) refine(acc, at, base[k - 1], (double)(k - 1) / intervals, base[k], (double)k / intervals, 0);
            push(base[k]);
        }
        if (failed) acc.failed = true;
    }
    // hands the samples to Accum as ONE polyline (pen up first), crossings located where the side changes
    void draw(Accum &acc, bool closed) {
        if (acc.failed || n == 0) return;
        acc.pen_up();
        for (int k = 0; k < n && !acc.failed; k++) {
            if (k > 0 && s[k].front != s[k - 1].front) acc.crossing(s[k - 1].u, s[k].u, s[k - 1].front);
            if (acc.failed) break;
            acc.feed(s[k].nd, s[k].front);
        }
        (void)closed;
    }
};

// margins relative to the part of the box that can matter (screens reach |u| <= aspect/2, |v| <= 1/2; clipped curves
// reach out to |u|, |v| ~ 25, which must not loosen the sides that lie on the screen; wider screens than 2.5 : 1 only
// get a margin that is smaller relative to their width, and the 1e-3 floor)
inline double clamp_u(double x) { return std::max(-1.25, std::min(1.25, x)); }     // screens up to 2.5 : 1
inline double clamp_v(double x) { return std::max(-0.55, std::min(0.55, x)); }
inline void margins(const Accum &acc, double rel_margin, double &mu, double &mv) {
    mu = rel_margin * (clamp_u(acc.u1) - clamp_u(acc.u0)) + 1.0e-3;
    mv = rel_margin * (clamp_v(acc.v1) - clamp_v(acc.v0)) + 1.0e-3;
}
// The adaptive sampling worked to a tolerance derived from an ESTIMATE of the on-screen extent; the margins come from the extent
// that was found.  If they turned out smaller than assumed (a large object of which a sliver is on the screen), the outline has to
// be followed again, to a quarter of the smaller margin.  Returns that tolerance, or 0 if the first pass was good enough.
inline double second_pass_tolerance(const Accum &acc, double rel_margin, double tol_used) {
    if (acc.failed || !acc.any_front) return 0.0;
    double mu, mv;
    margins(acc, rel_margin, mu, mv);
    const double need = 0.25 * std::min(mu, mv);
    return need < 0.999 * tol_used ? std::max(1.0e-4, need) : 0.0;
}

inline Rect finish(const Accum &acc, double rel_margin) {
    if (acc.failed) return full_rect();
    if (!acc.any_front) return acc.any_near_behind ? full_rect() : empty_rect();      // wholly (and well) behind the camera
    auto clu = [](double x) { return clamp_u(x); };
    auto clv = [](double x) { return clamp_v(x); };
    double mu, mv;
    margins(acc, rel_margin, mu, mv);
    const double r[4] = {acc.u0 - mu, acc.v0 - mv, acc.u1 + mu, acc.v1 + mv};
    for (double x : r) if (!std::isfinite(x)) return full_rect();
    auto clampf = [](double x) { return (float)std::max(-3.0e38, std::min(3.0e38, x)); };
    // round outwards
    Rect out = full_rect();
    out.u0 = std::nextafter(clampf(r[0]), -INFINITY); out.v0 = std::nextafter(clampf(r[1]), -INFINITY);
    out.u1 = std::nextafter(clampf(r[2]), INFINITY); out.v1 = std::nextafter(clampf(r[3]), INFINITY);
    // The diagonal slabs, if they cut enough off the box to pay for their test: a corner cut of size c (in u + v or u - v)
    // removes a triangle of area c^2 / 2 from the on-screen box; at least a quarter of it must go (a sphere's disc or a
    // box's hexagon loses 15-20 % to an octagon: not worth a second load and ten more instructions in every wave).
    if (acc.p_lo <= acc.p_hi) {
        const double U0 = clu(r[0]), U1 = clu(r[2]), V0 = clv(r[1]), V1 = clv(r[3]), md = mu + mv;
        const double plo = acc.p_lo - md, phi = acc.p_hi + md, mlo = acc.m_lo - md, mhi = acc.m_hi + md;
        auto sq = [](double c) { return c > 0.0 ? 0.5 * c * c : 0.0; };
        const double cut = sq((U1 + V1) - phi) + sq(plo - (U0 + V0)) + sq((U1 - V0) - mhi) + sq(mlo - (U0 - V1));
        if (cut >= 0.25 * (U1 - U0) * (V1 - V0) && std::isfinite(plo) && std::isfinite(phi) && std::isfinite(mlo) && std::isfinite(mhi)) {
            out.p_lo = std::nextafter((float)plo, -INFINITY); out.p_hi = std::nextafter((float)phi, INFINITY);
            out.m_lo = std::nextafter((float)mlo, -INFINITY); out.m_hi = std::nextafter((float)mhi, INFINITY);
        }
    }
    return out;
}

// does the ray from `oc` along `d` meet the box [lo, hi] (already inflated)?  Slab test; NaN -> true.
inline bool ray_meets_box(const double oc[3], D3 d, const double lo[3], const double hi[3]) {
    const double dd[3] = {d.x, d.y, d.z};
    double t0 = 0.0, t1 = 1.0e300;
    for (int a = 0; a < 3; a++) {
        if (dd[a] == 0.0) {
            if (oc[a] < lo[a] || oc[a] > hi[a]) return false;
            continue;
        }
        double ta = (lo[a] - oc[a]) / dd[a], tb = (hi[a] - oc[a]) / dd[a];
        if (ta > tb) std::swap(ta, tb);
        t0 = std::max(t0, ta);
        t1 = std::min(t1, tb);
    }
    return !(t0 > t1);
}

}  // namespace detail

// Box [bmin, bmax] in object space (cube: +-1; mesh: the root node's bounds).
inline Rect box_rect(const rpt_object &o, int interval, const double bmin[3], const double bmax[3]) {
    using namespace detail;
    DirMap m = make_map(o, interval);
    if (!m.ok) return full_rect();
    const D3 cam{o.stationaryCam.y, o.stationaryCam.z, o.stationaryCam.w};
    const D3 oc{o.InvM[0].x * cam.x + o.InvM[0].y * cam.y + o.InvM[0].z * cam.z + o.InvM[0].w,
                o.InvM[1].x * cam.x + o.InvM[1].y * cam.y + o.InvM[1].z * cam.z + o.InvM[1].w,
                o.InvM[2].x * cam.x + o.InvM[2].y * cam.y + o.InvM[2].z * cam.z + o.InvM[2].w};
    if (!finite3(oc)) return full_rect();
    const double lo[3] = {bmin[0], bmin[1], bmin[2]}, hi[3] = {bmax[0], bmax[1], bmax[2]}, p[3] = {oc.x, oc.y, oc.z};
    bool inside = true;
    double diag2 = 0.0;
    for (int a = 0; a < 3; a++) {
        if (!(hi[a] >= lo[a]) || !std::isfinite(lo[a]) || !std::isfinite(hi[a])) return full_rect();
        const double mrg = 0.05 * (hi[a] - lo[a]) + 1.0e-4;
        inside = inside && p[a] >= lo[a] - mrg && p[a] <= hi[a] + mrg;
        diag2 += (hi[a] - lo[a]) * (hi[a] - lo[a]);
    }
    if (inside) return full_rect();
    // the slab test's float error grows with the distance in box units: 4e-7 |oc| against the 5 % the box is inflated by
    if (dot(oc, oc) > 1.0e8 * std::max(1.0e-300, diag2)) return full_rect();
    // Which faces does oc see?  +1 seen, -1 hidden, 0 too close to the face's plane to say.  The outline of a convex box is
    // made of the edges between a seen and a hidden face; only those are sampled (edges between two seen or two hidden
    // faces lie inside the outline: taking or leaving them changes nothing).
    int seen_lo[3], seen_hi[3];
    for (int a = 0; a < 3; a++) {
        const double tol = 1.0e-6 * (hi[a] - lo[a]) + 1.0e-12;
        seen_lo[a] = p[a] < lo[a] - tol ? 1 : (p[a] > lo[a] + tol ? -1 : 0);
        seen_hi[a] = p[a] > hi[a] + tol ? 1 : (p[a] < hi[a] - tol ? -1 : 0);
    }
    // the eight corners are mapped once (corner k: bit 0/1/2 = x/y/z at hi); the first and the last one are pushed
    // through F again, which is where a wrong inverse would show
    D3 cu[8], cnd[8];
    for (int k = 0; k < 8; k++) {
        cu[k] = sub(D3{(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]}, oc);
        if (!m.G(cu[k], cnd[k])) return full_rect();
    }
    {   // the box's angular radius as seen from oc sets the tolerance of that check (and of every later one)
        double cmin = 1.0;
        const D3 centre = sub(D3{0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])}, oc);
        const double lc = len(centre);
        for (int k = 0; k < 8 && lc > 0.0; k++) {
            const double lk = len(cu[k]);
            if (lk > 0.0) cmin = std::min(cmin, dot(cu[k], centre) / (lk * lc));
        }
        const double ang = std::acos(std::max(-1.0, std::min(1.0, cmin)));
        if (!m.float_noise_small_against(ang)) return full_rect();
        m.tol = std::min(1.0e-4, 5.0e-3 * ang);
        if (!m.verify(cu[0], cnd[0]) || !m.verify(cu[7], cnd[7])) return full_rect();
    }
    // the on-screen extent of the box's corners: what the margin of the bounds — and with it the tolerance of the adaptive
    // sampling — is relative to (finish())
    const double rel_margin = m.linear ? 0.002 : 0.025;
    double extent = 0.0;
    {
        double a0 = 1e300, a1 = -1e300, b0 = 1e300, b1 = -1e300;
        for (int k = 0; k < 8; k++) {
            if (!Accum::is_front(cnd[k])) continue;
            const double iz = 0.5 / cnd[k].z, pu = std::max(-1.25, std::min(1.25, cnd[k].x * iz)), pv = std::max(-0.55, std::min(0.55, cnd[k].y * iz));
            a0 = std::min(a0, pu); a1 = std::max(a1, pu); b0 = std::min(b0, pv); b1 = std::max(b1, pv);
        }
        extent = a1 >= a0 ? std::min(a1 - a0, b1 - b0) : 0.0;            // the smaller one: each side's margin is relative to its own extent
        if (!(extent > 0.0)) extent = 1.0e-9;      // (no corner in front, or a box seen edge-on: the 1e-3 floor of the margin sets the tolerance)
    }
    double blo[3], bhi[3];
    for (int a = 0; a < 3; a++) { const double g = 0.05 * (hi[a] - lo[a]) + 1.0e-4; blo[a] = lo[a] - g; bhi[a] = hi[a] + g; }
    auto inside_box = [&](D3 d) { return !finite3(d) || ray_meets_box(p, d, blo, bhi); };
    static thread_local Curve curve;          // (640 samples: kept off the stack of every call)
    double tol_used = 0.0;
    auto outline = [&](Accum &acc, double forced_tol) {
    bool clipped = false;
    for (int axis = 0; axis < 3 && !acc.failed; axis++) {
        const int b = (axis + 1) % 3, c = (axis + 2) % 3;
        for (int corner = 0; corner < 4 && !acc.failed; corner++) {
            const int fb = (corner & 1) ? seen_hi[b] : seen_lo[b], fc = (corner & 2) ? seen_hi[c] : seen_lo[c];
            if (fb != 0 && fc != 0 && fb == fc) continue;      // both faces seen or both hidden: not on the outline
            double q[3];
            q[b] = (corner & 1) ? hi[b] : lo[b];
            q[c] = (corner & 2) ? hi[c] : lo[c];
            const int k_lo = ((corner & 1) ? (1 << b) : 0) | ((corner & 2) ? (1 << c) : 0), k_hi = k_lo | (1 << axis);
            auto at = [&](double t) {
                double r[3] = {q[0], q[1], q[2]};
                r[axis] = lo[axis] + (hi[axis] - lo[axis]) * t;
                return sub(D3{r[0], r[1], r[2]}, oc);
            };
            // straight edges stay straight under a linear map: its two corners are the edge — WHEN BOTH ARE IN FRONT of the clip
            // cone (the cone is convex).  If one or both are not, the stretch between them can still pass in front: the camera
            // direction runs linearly from corner to corner, and close to the camera its length shrinks faster than its z — a wall
            // a hundred units wide seen from a third of a unit away has all eight corners at the horizon, left and right, and fills
            // the screen in between (found by tools/verify_sweep.py, round 3: ladder_paradox.txt's back wall, 74 % of the frame
            // lost at one camera state in 16 000).  So such an edge is CUT with the cone exactly: nd(t) = A + t (B - A),
            // nd.z > EPS |nd| is a quadratic inequality in t, and the part of the edge in front is the segment between its roots.
            if (m.linear) {
                const D3 A = cnd[k_lo], B = cnd[k_hi];
                bool fa = false, fb2 = false;
                acc.classify(A, fa);
                acc.classify(B, fb2);
                acc.pen_up();
                if (fa && fb2) {
                    acc.point(A);
                    acc.point(B);
                } else {
                    clipped = true;
                    const D3 Dv = sub(B, A);
                    const double e2 = EPS_FRONT * EPS_FRONT;
                    const double qa = Dv.z * Dv.z - e2 * dot(Dv, Dv), qb = 2.0 * (A.z * Dv.z - e2 * dot(A, Dv)), qc = A.z * A.z - e2 * dot(A, A);
                    double cuts[4] = {0.0, 1.0, 0.0, 0.0};
                    int n_cuts = 2;
                    if (qa != 0.0) {
                        const double disc = qb * qb - 4.0 * qa * qc;
                        if (disc >= 0.0) {
                            const double sq = std::sqrt(disc), q = -0.5 * (qb + (qb >= 0.0 ? sq : -sq));
                            const double r1 = q / qa, r2 = q != 0.0 ? qc / q : r1;
                            if (r1 > 0.0 && r1 < 1.0) cuts[n_cuts++] = r1;
                            if (r2 > 0.0 && r2 < 1.0) cuts[n_cuts++] = r2;
                        }
                    } else if (qb != 0.0) {
                        const double r1 = -qc / qb;
                        if (r1 > 0.0 && r1 < 1.0) cuts[n_cuts++] = r1;
                    }
                    std::sort(cuts, cuts + n_cuts);
                    auto nd_at = [&](double t) { return add(A, mul(Dv, t)); };
                    double t_in0 = 2.0, t_in1 = -1.0;                    // the union of the pieces whose midpoint is in front (one piece: the cone is convex)
                    for (int c2 = 0; c2 + 1 < n_cuts; c2++) {
                        if (!(cuts[c2 + 1] > cuts[c2])) continue;
                        if (Accum::is_front(nd_at(0.5 * (cuts[c2] + cuts[c2 + 1])))) { t_in0 = std::min(t_in0, cuts[c2]); t_in1 = std::max(t_in1, cuts[c2 + 1]); }
                    }
                    if (t_in1 > t_in0) {
                        // (end points ON the cone have nd.z = EPS |nd| > 0: on the plane they lie on the horizon circle, where horizon() takes over)
                        const D3 P0 = nd_at(t_in0), P1 = nd_at(t_in1);
                        if (!(P0.z > 0.0) || !(P1.z > 0.0) || !finite3(P0) || !finite3(P1)) { acc.failed = true; break; }
                        acc.point(P0);
                        acc.point(P1);
                        acc.n_crossings += (t_in0 > 0.0) + (t_in1 < 1.0);
                    }
                }
                acc.pen_up();
                continue;
            }
            // Under aberration an edge becomes a conic arc, followed adaptively.  The BASE samples are uniform in the ANGLE the edge
            // subtends at the camera in the object's rest frame, not in the edge's own parameter: a beam 880 box-widths long that
            // passes half a unit from the camera has both ends far behind it and its middle tenth — or thousandth — in front, and
            // sixteenths of its LENGTH never land there (found by tools/verify_fuzz.py --kinds walls, round 3).  Seen from the
            // camera the edge is an arc of a great circle, at most 180 degrees; up to sixteen base intervals of at most 11.25 degrees,
            // each refined to a sixteenth where it is not wholly in front, resolve 0.7 degrees of it.
            const D3 R0 = m.to_rest(cu[k_lo]), R1 = m.to_rest(cu[k_hi]), Dr = sub(R1, R0);
            const double dd = dot(Dr, Dr);
            int n_base = 1;
            double s_star = 0.0, h_over_d = 0.0, th0 = 0.0, th1 = 0.0;
            bool by_angle = false;
            if (dd > 0.0 && finite3(R0) && finite3(R1)) {
                s_star = -dot(R0, Dr) / dd;
                const D3 foot = add(R0, mul(Dr, s_star));
                const double h = len(foot), ld = std::sqrt(dd);
                if (h > 1.0e-12 * ld && std::isfinite(h)) {
                    h_over_d = h / ld;
                    th0 = std::atan((0.0 - s_star) / h_over_d);
                    th1 = std::atan((1.0 - s_star) / h_over_d);
                    n_base = std::max(1, std::min(16, (int)std::ceil(std::fabs(th1 - th0) / (M_PI / 16.0))));
                    by_angle = n_base > 1;
                }
            }
            auto at_angle = [&](double tau) {
                const double sp = by_angle ? std::min(1.0, std::max(0.0, s_star + h_over_d * std::tan(th0 + tau * (th1 - th0)))) : tau;
                return at(sp);
            };
            curve.sample(acc, at_angle, n_base, true, &cu[k_lo], &cnd[k_lo], &cu[k_hi], &cnd[k_hi], rel_margin, extent, forced_tol);
            tol_used = std::max(tol_used, curve.tol);
            clipped = clipped || curve.clipped;
            curve.draw(acc, false);                             // every edge is a polyline of its own
        }
    }
    if (acc.failed) return;
    // the horizon: when the whole outline is in front of the camera only a region that wraps around behind the camera can
    // reach it (strong aberration can produce one; wraps_behind asks); always when the outline is clipped
    if (!clipped) {
        // (a linear map keeps the cone of kept directions convex: with its whole outline in front of the camera it cannot
        // reach the horizon, so only aberrated outlines are asked)
        if (!m.linear && acc.wraps_behind(inside_box)) acc.horizon(32, inside_box);
    } else {
        acc.horizon(32, inside_box);
    }
    if (acc.reaches_window_edge()) acc.window_corners(inside_box);
    };
    Accum acc(m);
    outline(acc, -1.0);
    if (!m.linear) {
        const double again = second_pass_tolerance(acc, rel_margin, tol_used);
        if (again > 0.0) {
            Accum finer(m);
            tol_used = 0.0;
            outline(finer, again);
            acc = finer;
        }
    }
    return finish(acc, rel_margin);
}

// Unit sphere at the object-space origin.
inline Rect sphere_rect(const rpt_object &o, int interval) {
    using namespace detail;
    DirMap m = make_map(o, interval);
    if (!m.ok) return full_rect();
    const D3 cam{o.stationaryCam.y, o.stationaryCam.z, o.stationaryCam.w};
    const D3 oc{o.InvM[0].x * cam.x + o.InvM[0].y * cam.y + o.InvM[0].z * cam.z + o.InvM[0].w,
                o.InvM[1].x * cam.x + o.InvM[1].y * cam.y + o.InvM[1].z * cam.z + o.InvM[1].w,
                o.InvM[2].x * cam.x + o.InvM[2].y * cam.y + o.InvM[2].z * cam.z + o.InvM[2].w};
    const double dist = len(oc);
    // inflated radius.  The kernel decides a hit by the sign of b^2 - c in float with c = |oc|^2 - 1 (opencl_kernel.cl:341-345):
    // both terms are of size |oc|^2 and each carries a few ulp, so the discriminant is off by up to ~1.5e-6 |oc|^2 — for a
    // camera that is hundreds of radii away in the object's frame (a strongly boosted sphere) the float sphere is visibly
    // larger or smaller than the exact one.  The rim is taken from a sphere that is larger by that bound.
    if (!finite3(oc) || !std::isfinite(dist)) return full_rect();
    // (2^-19 (|oc|^2 + 1): the bound rpt_bounds_certify.hpp DERIVES for that discriminant — the claim made here has to be provable there)
    const double rb = 1.02 * std::sqrt(1.0 + 1.9073486328125e-6 * (dist * dist + 1.0));
    if (!(dist > 1.1 * rb) || dist > 5.0e4) return full_rect();      // (beyond 5e4 radii the float direction itself is too coarse)
    const D3 axis = mul(oc, -1.0 / dist);
    const double sa = rb / dist, ca = std::sqrt(1.0 - sa * sa);
    if (!m.float_noise_small_against(sa)) return full_rect();
    m.tol = std::min(1.0e-4, 5.0e-3 * sa);
    const D3 helper = std::fabs(axis.x) < 0.6 ? D3{1, 0, 0} : D3{0, 1, 0};
    D3 e1 = cross(axis, helper);
    e1 = mul(e1, 1.0 / len(e1));
    const D3 e2 = cross(axis, e1);
    // the rim of the tangent cone, followed adaptively from twelve base intervals (a circle seen under perspective or aberration
    // is anything but uniformly parametrised by its own angle; an undistorted one is done with these and their midpoints)
    auto at = [&](double t) {
        const double phi = 2.0 * M_PI * t;
        return add(mul(axis, ca), mul(add(mul(e1, std::cos(phi)), mul(e2, std::sin(phi))), sa));
    };
    static thread_local Curve curve;
    const D3 u_first = at(0.0);
    D3 nd_first;
    if (!m.G_checked(u_first, nd_first)) return full_rect();
    const double cos_in = std::cos(std::min(1.5, std::asin(sa) * 1.05 + 0.01));      // the cone, a little wider
    auto inside_cone = [&](D3 d) {
        const double l = len(d);
        return !finite3(d) || !(l > 0.0) || dot(d, axis) >= cos_in * l;
    };
    double tol_used = 0.0;
    auto outline = [&](Accum &acc, double forced_tol) {
        curve.sample(acc, at, 12, true, &u_first, &nd_first, &u_first, &nd_first, 0.04, 0.0, forced_tol);      // closed: ends where it began
        tol_used = curve.tol;
        const bool clipped = curve.clipped;
        // every eighth sample is pushed through F again (a wrong inverse shows there too)
        for (int k = 0; k < curve.n && !acc.failed; k += 8)
            if (!m.verify(curve.s[k].u, curve.s[k].nd)) acc.failed = true;
        curve.draw(acc, true);
        if (acc.failed) return;
        if (!clipped) {
            if (!m.linear && acc.wraps_behind(inside_cone)) acc.horizon(32, inside_cone);
        } else {
            acc.horizon(32, inside_cone);
        }
        if (acc.reaches_window_edge()) acc.window_corners(inside_cone);
    };
    Accum acc(m);
    outline(acc, -1.0);
    const double again = second_pass_tolerance(acc, 0.04, tol_used);
    if (again > 0.0) {
        Accum finer(m);
        outline(finer, again);
        acc = finer;
    }
    return finish(acc, 0.04);
}

// The rectangle of one object (type-dispatched).  root_bounds: min.xyz,max.xyz of a mesh object's root node, or null.
inline Rect object_rect(const rpt_object &o, int interval, const float *root_bounds) {
    if (o.type == RPT_SPHERE) return sphere_rect(o, interval);
    if (o.type == RPT_CUBE) {
        const double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
        return box_rect(o, interval, lo, hi);
    }
    if (o.type == RPT_MESH && root_bounds) {
        const double lo[3] = {root_bounds[0], root_bounds[1], root_bounds[2]}, hi[3] = {root_bounds[3], root_bounds[4], root_bounds[5]};
        return box_rect(o, interval, lo, hi);
    }
    return full_rect();
}

}  // namespace rptb
