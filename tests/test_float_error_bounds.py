"""The float-error bounds the culls rest on (csrc/rpt_bounds_certify.hpp section 2, and the shadow-segment culls of
csrc/rpt_kernels.hip.h::intersect_object), checked against the oracle's fp32 intersectors on adversarial rays: far origins
(the sphere's discriminant loses its digits), grazing rays, rays that just miss.  A derivation can have a slip; a numerical
counter-example would show it.  Host only."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi
from conftest import load_config

U = 2.0 ** -24
f32 = np.float32


def _unit_object(scene, kind):
    """One object of the given type (0 sphere, 1 cube) with identity M / InvM / Lorentz: rest frame = object space."""
    o = scene.objects()[:1].copy()
    eye = np.eye(4, dtype=np.float32)
    for f in ("M", "InvM", "Lorentz", "InvLorentz"):
        o[f][0] = eye.reshape(o[f][0].shape)
    o["type"] = kind
    o["textureIndex"] = -1
    o["light"] = 0
    return o


def _rays(rng, n, far):
    """Origins at |o| from 1.2 to `far` (log-uniform), directions aimed at points around the unit shape's silhouette."""
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = np.exp(rng.uniform(np.log(1.2), np.log(far), size=n))
    o = d * dist[:, None]
    target = rng.normal(size=(n, 3))
    target /= np.linalg.norm(target, axis=1, keepdims=True)
    target *= rng.choice([0.97, 0.999, 1.0, 1.0, 1.001, 1.03, 1.4, 1.75], size=n)[:, None] * rng.choice([1.0, 1.7320508], size=n)[:, None] ** rng.integers(0, 2, size=n)[:, None]
    g = target - o
    g *= np.exp(rng.uniform(-3, 3, size=n))[:, None]            # any length: the intersector normalises
    return o.astype(np.float32), g.astype(np.float32)


def _normalised_as_the_kernel_does(g):
    g = g.astype(np.float32)
    dot = (g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]
    scale = np.sqrt(dot, dtype=np.float32)
    return (g / scale[:, None]).astype(np.float32), scale


@pytest.mark.parametrize("far", [30.0, 3.0e3, 4.0e4])
def test_a_float_sphere_hit_is_an_exact_hit_of_the_inflated_sphere(far):
    scene = load_config("shadows")
    obj = _unit_object(scene, 0)
    rng = np.random.default_rng(int(far))
    o, g = _rays(rng, 400000, far)
    rays = np.concatenate([np.zeros((len(o), 1), np.float32), o, np.zeros((len(o), 1), np.float32), g], axis=1)
    out = oracle_ffi.object_rays(scene, 0, 0, rays, objects=obj)
    hit = out[:, 0] > 0
    assert hit.sum() > 10000
    gn, scale = _normalised_as_the_kernel_does(g)
    o64, g64 = o.astype(np.float64), gn.astype(np.float64)
    oo = (o64 * o64).sum(axis=1)
    b = -(o64 * g64).sum(axis=1)
    gg = (g64 * g64).sum(axis=1)
    line2 = oo - b * b / gg                                            # squared distance of the exact line from the centre
    R2 = 1.0 + 32.0 * U * (oo + 1.0)                                   # rpt_bounds_certify.hpp section 2
    assert (b[hit] > 0).all(), "a hit on a ray that points away from the sphere"
    worst = np.max((line2[hit] - 1.0) / (oo[hit] + 1.0)) / U
    assert (line2[hit] <= R2[hit]).all(), f"a float hit whose exact line passes outside R': excess {worst:.1f} u (|o|^2 + 1), bound 32"
    # the hit POINT (shadow-segment cull): o + g * dist with dist = hit.dist * scale lies within [-m, m]^3, m = 1 + 2^-19 (1 + |o|^2)
    dist = out[:, 1].astype(np.float64) * scale.astype(np.float64)
    P = o64 + g64 * dist[:, None]
    m = 1.0 + 2.0 ** -19 * (1.0 + oo)
    rel = (np.abs(P[hit]).max(axis=1) - 1.0) / (1.0 + oo[hit]) / U
    assert (np.abs(P[hit]).max(axis=1) <= m[hit]).all(), f"hit point outside the cull's box: excess {rel.max():.1f} u (1 + |o|^2), bound 32"
    print(f"far {far:g}: {hit.sum()} hits; line excess {worst:.2f} u (|o|^2 + 1) of 32; hit point excess {rel.max():.2f} u (1 + |o|^2) of 32")


@pytest.mark.parametrize("far", [30.0, 3.0e3, 1.0e5])
def test_a_float_cube_hit_is_an_exact_hit_of_the_inflated_cube(far):
    scene = load_config("shadows")
    obj = _unit_object(scene, 1)
    rng = np.random.default_rng(int(far) + 1)
    o, g = _rays(rng, 400000, far)
    keep = np.abs(o).max(axis=1) > 1.001                               # origins outside the cube (winding +1), as the certificate requires
    o, g = o[keep], g[keep]
    rays = np.concatenate([np.zeros((len(o), 1), np.float32), o, np.zeros((len(o), 1), np.float32), g], axis=1)
    out = oracle_ffi.object_rays(scene, 0, 0, rays, objects=obj)
    hit = out[:, 0] > 0
    assert hit.sum() > 10000
    gn, scale = _normalised_as_the_kernel_does(g)
    o64, g64 = o.astype(np.float64), gn.astype(np.float64)
    dist = out[:, 1].astype(np.float64) * scale.astype(np.float64)     # (hit.dist = dist / scale: one rounding, 1 u of the parameter)
    assert (dist[hit] >= 0).all()
    P = o64 + g64 * dist[:, None]
    omax = np.abs(o64).max(axis=1)
    m = 1.0 + 4.0 * U * (1.0 + omax)
    # the parameter is known to 1 u here (the division by scale), which moves the point by |g dist| u <= (|o| + 2) u: allow for it
    slack = 1.5 * U * (omax + 2.0)
    rel = (np.abs(P[hit]).max(axis=1) - 1.0) / (1.0 + omax[hit]) / U
    assert (np.abs(P[hit]).max(axis=1) <= m[hit] + slack[hit]).all(), f"cube hit point outside [-m, m]^3: excess {rel.max():.2f} u (1 + |o|), bound 4 (+1.5 for this test's own rounding)"
    print(f"far {far:g}: {hit.sum()} hits; hit point excess {rel.max():.2f} u (1 + |o|max) of 4 (+1.5)")


def test_a_true_root_box_test_is_an_exact_hit_of_the_grown_box():
    rng = np.random.default_rng(5)
    lib = oracle_ffi.lib()
    FP = C.POINTER(C.c_float)
    worst = 0.0
    passed = 0
    for trial in range(60000):
        c = rng.normal(scale=rng.choice([0.1, 1.0, 50.0]), size=3)
        h = np.exp(rng.uniform(-4, 2, size=3))
        lo, hi = (c - h).astype(np.float32), (c + h).astype(np.float32)
        o = (c + rng.normal(size=3) * h * rng.choice([1.5, 5.0, 300.0, 2.0e4])).astype(np.float32)
        corner = np.where(rng.random(3) < 0.5, lo, hi).astype(np.float64)
        edge_pt = corner + (rng.random(3) < 0.4) * rng.uniform(-1, 1, size=3) * h * 0.5
        g = (edge_pt + rng.normal(scale=rng.choice([0.0, 1e-7, 1e-4, 1e-2]), size=3) * h) - o
        if rng.random() < 0.2:
            g[rng.integers(0, 3)] = 0.0                                 # axis-parallel rays: infinite / NaN plane distances
        gn = _normalised_as_the_kernel_does(g[None, :].astype(np.float32))[0][0]
        if not np.isfinite(gn).all():
            continue
        d2 = (C.c_float * 2)()
        s2 = (C.c_int * 2)()
        ok = lib.rpt_oracle_aabb(lo.ctypes.data_as(FP), hi.ctypes.data_as(FP), o.ctypes.data_as(FP), gn.ctypes.data_as(FP), d2, s2)
        if not ok:
            continue
        passed += 1
        o64, g64, lo64, hi64 = o.astype(np.float64), gn.astype(np.float64), lo.astype(np.float64), hi.astype(np.float64)
        grow = 4.0 * U * (np.maximum(np.abs(lo64), np.abs(hi64)) + np.abs(o64)) + 1e-30
        t0, t1 = 0.0, np.inf
        for k in range(3):
            a, b = lo64[k] - grow[k], hi64[k] + grow[k]
            if g64[k] == 0.0:
                if o64[k] < a or o64[k] > b:
                    t1 = -1.0
                continue
            ta, tb = (a - o64[k]) / g64[k], (b - o64[k]) / g64[k]
            t0, t1 = max(t0, min(ta, tb)), min(t1, max(ta, tb))
        assert t0 <= t1, f"trial {trial}: the float slab test passes, the exact forward ray misses the grown box (lo {lo}, hi {hi}, o {o}, g {gn})"
        # how much of the growth was needed?  (exact ray against the un-grown box, distance of the miss in units of the growth)
    assert passed > 5000


def _misses_root_as_the_kernel_does(c, mh, o, D):
    """mesh_ray_misses_root of csrc/rpt_kernels.hip.h, operation for operation in float32 (vectorised over rays)."""
    c, mh, o, D = (x.astype(np.float32) for x in (c, mh, o, D))
    p = c - o
    h = mh + f32(6.0e-7) * np.abs(o)
    miss = np.zeros(len(o), dtype=bool)
    for k in range(3):
        miss |= ((p[:, k] < -h[:, k]) & (D[:, k] >= 0)) | ((p[:, k] > h[:, k]) & (D[:, k] <= 0))
    a = np.abs(D)
    for (j, k) in ((1, 2), (2, 0), (0, 1)):
        m1, m2 = p[:, j] * D[:, k], p[:, k] * D[:, j]
        miss |= np.abs(m1 - m2) > (h[:, j] * a[:, k] + h[:, k] * a[:, j]) * f32(1.000001) + f32(5.0e-7) * (np.abs(m1) + np.abs(m2))
    return miss


def test_rays_the_kernel_skips_fail_the_float_root_test():
    """mesh_ray_misses_root (restated in numpy float32) true  =>  the oracle's intersect_AABB on the NORMALISED ray is false.
    Rays aimed at and just past the box's corners, edges and faces, from near and far, axis-parallel ones included; and the
    predicate must be useful: it fires for most rays that do miss."""
    rng = np.random.default_rng(11)
    lib = oracle_ffi.lib()
    FP = C.POINTER(C.c_float)
    n_skip = n_miss = n_total = 0
    for trial in range(300):
        cen = rng.normal(scale=rng.choice([0.1, 1.0, 50.0]), size=3)
        half = np.exp(rng.uniform(-4, 2, size=3))
        lo, hi = (cen - half).astype(np.float32), (cen + half).astype(np.float32)
        c = (f32(0.5) * (lo + hi)).astype(np.float32)                       # the centre as build_dobjs forms it
        hext = np.maximum(hi.astype(np.float64) - c, c.astype(np.float64) - lo)
        mh = np.nextafter((hext + 8.0 * U * np.maximum(np.abs(lo), np.abs(hi)) + 1e-30).astype(np.float32), f32(np.inf))
        n = 400
        o = (cen + rng.normal(size=(n, 3)) * half * rng.choice([1.5, 5.0, 300.0, 2.0e4], size=(n, 1))).astype(np.float32)
        corner = np.where(rng.random((n, 3)) < 0.5, lo, hi).astype(np.float64)
        target = corner + (rng.random((n, 3)) < 0.4) * rng.uniform(-1, 1, size=(n, 3)) * half * 0.5
        target += rng.normal(size=(n, 3)) * half * rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-3, 0.1], size=(n, 1))
        D = (target - o) * np.exp(rng.uniform(-3, 3, size=(n, 1)))
        D[rng.random(n) < 0.15, rng.integers(0, 3)] = 0.0
        D = D.astype(np.float32)
        skip = _misses_root_as_the_kernel_does(np.tile(c, (n, 1)), np.tile(mh, (n, 1)), o, D)
        g, _ = _normalised_as_the_kernel_does(D)
        for i in range(n):
            if not np.isfinite(g[i]).all():
                continue
            d2 = (C.c_float * 2)()
            s2 = (C.c_int * 2)()
            passes = lib.rpt_oracle_aabb(lo.ctypes.data_as(FP), hi.ctypes.data_as(FP), o[i].ctypes.data_as(FP), g[i].ctypes.data_as(FP), d2, s2)
            n_total += 1
            n_miss += not passes
            n_skip += bool(skip[i])
            assert not (skip[i] and passes), f"trial {trial} ray {i}: skipped, but the float root test passes (lo {lo}, hi {hi}, o {o[i]}, D {D[i]})"
    assert n_skip >= 0.8 * n_miss > 0, (n_skip, n_miss, n_total)
    print(f"{n_total} rays: {n_miss} fail the float root test, {n_skip} of them are skipped without the normalisation")


def test_the_primary_direction_error_budget():
    """rpt_bounds_certify.hpp section 1: the object-space direction the kernel hands its intersectors differs from the exact
    A (interval, d / |d|) by at most ERR_i = 24u sum_k |InvM_ik| sum_j |Lorentz_kj| per component (as a direction).  The kernel's
    chain — the camera direction normalised TWICE, the 4-term dot with Lorentz, the 3-term dot with InvM, the normalisation — is
    restated here in numpy float32, operation for operation, on the objects of shipped and generated scenes at random points of the
    window, and compared with the same chain in float64."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import verify_fuzz
    rng = np.random.default_rng(21)
    scenes = [load_config(n) for n in ("shadows", "cubes", "arch", "bunny")] + [verify_fuzz.build(k, 7000 + i)[0] for i, k in enumerate(("random", "extreme", "close", "walls", "ellipsoids", "meshwalls") * 6)]
    worst = 0.0
    checked = 0
    for s in scenes:
        interval = np.float32(s.params["interval"])
        for o in s.objects()[:64]:
            L = np.array(o["Lorentz"], dtype=np.float32).reshape(4, 4)
            Iv = np.array(o["InvM"], dtype=np.float32).reshape(4, 4)
            if not (np.isfinite(L).all() and np.isfinite(Iv).all()):
                continue
            n = 2000
            fu = rng.uniform(-2, 2, size=n).astype(np.float32)
            fv = rng.uniform(-0.5, 0.5, size=n).astype(np.float32)
            d = np.stack([fu, fv, np.full(n, 0.5, np.float32)], axis=1)

            def norm32(v):
                dot = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
                return (v / np.sqrt(dot, dtype=np.float32)[:, None]).astype(np.float32)
            nd = norm32(norm32(d))
            w = np.concatenate([np.full((n, 1), interval, np.float32), nd], axis=1)
            d3 = np.stack([((L[k, 0] * w[:, 0] + L[k, 1] * w[:, 1]) + L[k, 2] * w[:, 2]) + L[k, 3] * w[:, 3] for k in (1, 2, 3)], axis=1).astype(np.float32)
            dirf = np.stack([(Iv[i, 0] * d3[:, 0] + Iv[i, 1] * d3[:, 1]) + Iv[i, 2] * d3[:, 2] for i in range(3)], axis=1).astype(np.float32)
            scale = np.sqrt((dirf[:, 0] * dirf[:, 0] + dirf[:, 1] * dirf[:, 1]) + dirf[:, 2] * dirf[:, 2], dtype=np.float32)
            dirn = (dirf / scale[:, None]).astype(np.float32)
            # exact (float64 on the float matrices)
            d64 = d.astype(np.float64)
            nd64 = d64 / np.linalg.norm(d64, axis=1, keepdims=True)
            w64 = np.concatenate([np.full((n, 1), float(interval)), nd64], axis=1)
            A = Iv[:3, :3].astype(np.float64) @ L[1:4, :].astype(np.float64)
            D = w64 @ A.T
            S = np.abs(Iv[:3, :3].astype(np.float64)) @ np.abs(L[1:4, :].astype(np.float64)).sum(axis=1)
            if not (np.isfinite(D).all() and np.isfinite(dirn).all() and (S > 0).all()):
                continue
            # as a direction: dirn * (exact length of the float dir) against D
            lenf = np.linalg.norm(dirf.astype(np.float64), axis=1, keepdims=True)
            err = np.abs(dirn.astype(np.float64) * lenf - D) / S[None, :]
            worst = max(worst, float(err.max()) / U)
            checked += n
    assert checked > 100000
    assert worst <= 24.0, f"direction error {worst:.1f} u of the row sums: above the certificate's budget of 24"
    print(f"{checked} directions: worst component error {worst:.2f} u x sum_k |InvM_ik| sum_j |L_kj| (budget 24)")
