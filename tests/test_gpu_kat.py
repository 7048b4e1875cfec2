"""Known-answer tests of single device functions (rpt_probe) against the oracle's per-function entry
points, on random and edge-case inputs: bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def renderer():
    from relativitypathtracer_amd.renderer import Renderer
    r = Renderer(0)
    yield r
    r.close()


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_intersect_triangle(renderer):
    rng = np.random.default_rng(7)
    n = 4096
    tri = rng.uniform(-2, 2, size=(n, 9)).astype(np.float32)
    org = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    # aim most rays at a random point of their triangle so that a good share hits
    w = rng.dirichlet((1, 1, 1), size=n).astype(np.float32)
    target = (tri.reshape(n, 3, 3) * w[:, :, None]).sum(axis=1)
    d = target - org + rng.normal(0, 0.05, size=(n, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # edge cases: ray in the triangle's plane, degenerate triangle, axis-parallel rays, ray through a vertex
    tri[0] = [0, 0, 0, 1, 0, 0, 0, 1, 0]; org[0] = [-1, 0.2, 0]; d[0] = [1, 0, 0]
    tri[1] = [1, 1, 1, 1, 1, 1, 2, 2, 2]
    tri[2] = [0, 0, 1, 1, 0, 1, 0, 1, 1]; org[2] = [0, 0, 0]; d[2] = [0, 0, 1]
    tri[3] = [0, 0, 1, 1, 0, 1, 0, 1, 1]; org[3] = [0.25, 0.25, 0]; d[3] = [0, 0, 1]
    tri[4] = [0, 0, 1, 1, 0, 1, 0, 1, 1]; org[4] = [0.25, 0.25, 2]; d[4] = [0, 0, 1]     # behind the origin: negative dist
    inp = np.concatenate([tri, org, d], axis=1)
    got = renderer.probe(0, inp, 4)
    lib = oracle_ffi.lib()
    want = np.zeros((n, 4), np.float32)
    for i in range(n):
        o3 = np.zeros(3, np.float32)
        h = lib.rpt_oracle_tri(_fp(tri[i, 0:3].copy()), _fp(tri[i, 3:6].copy()), _fp(tri[i, 6:9].copy()), _fp(org[i].copy()), _fp(d[i].copy()), _fp(o3))
        want[i] = [h, *o3]
    assert want[:, 0].sum() > n // 4
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_intersect_aabb(renderer):
    rng = np.random.default_rng(11)
    n = 4096
    lo = rng.uniform(-2, 0, size=(n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0.1, 3, size=(n, 3))).astype(np.float32)
    org = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    aim = ((lo + hi) / 2 - org + rng.normal(0, 0.8, size=(n, 3))).astype(np.float32)
    d[: n // 2] = aim[: n // 2]                                          # half the rays aimed near the box
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    org[:200] = ((lo[:200] + hi[:200]) / 2).astype(np.float32)          # origin inside the box
    d[200:260] = [1, 0, 0]; d[260:320] = [0, -1, 0]; d[320:380] = [0, 0, 1]    # axis-parallel: 1/0 = inf, 0*inf = NaN
    d[380:400, 0] = -0.0
    org[400:420, 0] = lo[400:420, 0]                                     # origin exactly on a face plane
    inp = np.concatenate([lo, hi, org, d], axis=1)
    got = renderer.probe(1, inp, 5)
    lib = oracle_ffi.lib()
    want = np.zeros((n, 5), np.float32)
    for i in range(n):
        d2 = np.zeros(2, np.float32); s2 = (C.c_int * 2)()
        h = lib.rpt_oracle_aabb(_fp(lo[i].copy()), _fp(hi[i].copy()), _fp(org[i].copy()), _fp(d[i].copy()), _fp(d2), s2)
        want[i] = [h, d2[0], d2[1], s2[0], s2[1]]
    assert 0.2 < want[:, 0].mean() < 0.9
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_camera_ray_every_column_and_row(renderer):
    lib = oracle_ffi.lib()
    cases = []
    for (W, H) in [(640, 480), (1920, 1080), (3840, 2160), (7680, 4320), (333, 77)]:
        xs = np.unique(np.concatenate([np.arange(0, W, max(W // 97, 1)), [W - 1]]))
        ys = np.unique(np.concatenate([np.arange(0, H, max(H // 89, 1)), [H - 1]]))
        for x in xs:
            for y in ys[:: max(len(ys) // 9, 1)]:
                cases.append((x, y, W, H))
    inp = np.array(cases, np.float32)
    got = renderer.probe(2, inp, 3)
    want = np.zeros_like(got)
    o = np.zeros(3, np.float32)
    for i, (x, y, W, H) in enumerate(cases):
        lib.rpt_oracle_camray(float(x), float(y), int(W), int(H), _fp(o))
        want[i] = o
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_hable(renderer):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(0, 30, size=(3000, 3)), rng.uniform(0, 1e-3, size=(500, 3)),
                        np.array([[0, 0, 0], [1, 1, 1], [0.15, 0.15, 0.25], [10, 10, 10], [2, 2, 2], [1e20, 1e-30, 5]])]).astype(np.float32)
    got = renderer.probe(3, x, 3)
    lib = oracle_ffi.lib()
    want = np.zeros_like(got)
    o = np.zeros(3, np.float32)
    for i in range(len(x)):
        lib.rpt_oracle_hable(_fp(x[i].copy()), _fp(o))
        want[i] = o
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_asin_atan2_device_equals_oracle(renderer):
    """The textured-sphere (u,v) functions: device and oracle carry the same explicit algorithm, so they agree bit
    for bit — on random, tiny, huge, signed-zero, infinite and NaN arguments."""
    lib = oracle_ffi.lib()
    rng = np.random.default_rng(21)
    n = 20000
    x = np.empty((n, 3), dtype=np.float32)
    x[:, 0] = rng.uniform(-1.05, 1.05, n)
    x[:, 1:] = rng.normal(size=(n, 2)) * rng.choice([1e-30, 1e-6, 1.0, 1e4, 1e30], size=(n, 1))
    specials = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 2.0 ** -13, 3e38, 1e-45]
    k = 0
    for a in specials:
        for b in specials:
            x[k] = [a if abs(a) <= 1 or not np.isfinite(a) else 0.3, a, b]
            k += 1
    got = renderer.probe(4, x, 2)
    want = np.empty((n, 2), dtype=np.float32)
    o = np.zeros(2, dtype=np.float32)
    for i in range(n):
        lib.rpt_oracle_asin_atan2(float(x[i, 0]), float(x[i, 1]), float(x[i, 2]), _fp(o))
        want[i] = o
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), x[~same.all(axis=1)][:5]


def test_walk_steps_device_equals_oracle(renderer):
    """getOppositeBoxSide (opencl_kernel.cl:172-198) with the hoisted reciprocals, and the octree child step
    (:237-238) in its general form and in the kernel's exact fast form for 0 <= uv < 1.5: bit-identical to the
    oracle on points inside, on and just outside the unit cell, axis-parallel directions, NaN."""
    lib = oracle_ffi.lib()
    rng = np.random.default_rng(33)
    n = 20000
    x = np.empty((n, 6), dtype=np.float32)
    d = rng.normal(size=(n, 3))
    x[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
    x[:, 3:] = rng.uniform(-0.1, 1.6, size=(n, 3))
    x[:2000, 3:] = rng.choice([0.0, 0.5, 1.0, 0.49999997, 0.99999994, 1.0000001, 1.4999999, 1.5], size=(2000, 3))
    x[2000:2300, rng.integers(0, 3)] = 0.0            # axis-parallel: an infinite reciprocal
    x[2300:2310, 3] = np.nan
    got = renderer.probe(5, x, 12)
    a, b = np.zeros(4, dtype=np.float32), np.zeros(4, dtype=np.float32)
    for i in range(n):
        lib.rpt_oracle_walk_steps(_fp(x[i, :3].copy()), _fp(x[i, 3:].copy()), _fp(a), _fp(b))
        for lo, want in ((0, a), (4, b), (8, b)):
            g = got[i, lo:lo + 4]
            same = (g.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(want))
            assert same.all(), (i, lo, x[i], g, want)


@pytest.mark.parametrize("scene_name", ["bunny", "shadows"])
def test_octree_walk_at_ray_level(scene_name):
    """intersect_octree (opencl_kernel.cl:200-308) ray by ray instead of frame by frame: the three walks of the product library —
    the reference's layouts, the throughput walk (kernel 41) and the latency walk (kernel 43: records ahead, a leaf's first record
    with its node) — and the oracle, on rays a camera never shoots: origins inside the root box and inside leaves, on faces, edges
    and corners of the box, directions parallel to one and two axes, rays aimed at mesh vertices (shared by many triangles and,
    where they lie on cell boundaries, by many leaves), rays that graze the box, rays that miss.  Hit flag, re-measured
    distance, interpolated normal and texture coordinates bit for bit."""
    from relativitypathtracer_amd import Scene
    from relativitypathtracer_amd.renderer import Renderer
    scene = Scene.from_file(scene_name)
    scene.update_objects()
    objs = scene.objects()
    mesh = int(np.flatnonzero(np.asarray(objs["type"]) == 2)[0])
    buf = scene.buffers()
    root = np.ascontiguousarray(buf["octrees"]).view(np.float32).reshape(-1, 24)[int(objs["meshIndex"][mesh])]     # rpt_octree: min.xyzw, max.xyzw, ...
    lo, hi = root[0:3].astype(np.float64), root[4:7].astype(np.float64)
    verts = np.asarray(buf["vertices"])[:, :3].astype(np.float64)
    rng = np.random.default_rng(20261005)
    n = 60000
    rays = np.empty((n, 6), dtype=np.float64)
    unit = lambda v: v / np.linalg.norm(v, axis=-1, keepdims=True)                                     # noqa: E731
    # 0: from outside towards points of the box;  1: from inside the box, any direction;  2: towards mesh vertices
    k = n // 6
    far = unit(rng.normal(size=(k, 3))) * rng.uniform(1.5, 6.0, size=(k, 1)) * np.linalg.norm(hi - lo) + 0.5 * (lo + hi)
    rays[:k, :3] = far
    rays[:k, 3:] = unit(rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), size=(k, 3)) - far)
    rays[k:2 * k, :3] = rng.uniform(lo, hi, size=(k, 3))
    rays[k:2 * k, 3:] = unit(rng.normal(size=(k, 3)))
    o2 = unit(rng.normal(size=(k, 3))) * 3.0 * np.linalg.norm(hi - lo) + 0.5 * (lo + hi)
    rays[2 * k:3 * k, :3] = o2
    rays[2 * k:3 * k, 3:] = unit(verts[rng.integers(0, len(verts), size=k)] - o2)
    # 3: axis-parallel (one or two zero components), from outside and inside
    d3 = np.zeros((k, 3))
    ax = rng.integers(0, 3, size=k)
    d3[np.arange(k), ax] = rng.choice([-1.0, 1.0], size=k)
    two = rng.random(k) < 0.5
    d3[two, (ax[two] + 1) % 3] = rng.normal(size=int(two.sum()))
    rays[3 * k:4 * k, 3:] = unit(d3)
    rays[3 * k:4 * k, :3] = rng.uniform(lo - 0.5 * (hi - lo), hi + 0.5 * (hi - lo), size=(k, 3))
    # 4: origins ON the box (faces, edges, corners), directions inwards and along the faces
    ob = rng.uniform(lo, hi, size=(k, 3))
    snap = rng.random((k, 3)) < 0.5
    snap[np.arange(k), rng.integers(0, 3, size=k)] = True
    side = rng.random((k, 3)) < 0.5
    ob = np.where(snap, np.where(side, lo, hi), ob)
    rays[4 * k:5 * k, :3] = ob
    rays[4 * k:5 * k, 3:] = unit(rng.normal(size=(k, 3)))
    # 5: grazing: along a face at a hair's distance, and plain misses
    rest = n - 5 * k
    og = rng.uniform(lo - 1.0 * (hi - lo), hi + 1.0 * (hi - lo), size=(rest, 3))
    axis = rng.integers(0, 3, size=rest)
    og[np.arange(rest), axis] = np.where(rng.random(rest) < 0.5, lo[axis], hi[axis]) * (1.0 + rng.choice([-1e-6, 0.0, 1e-6], size=rest))
    dg = rng.normal(size=(rest, 3))
    dg[np.arange(rest), axis] *= rng.choice([0.0, 1e-4, 1.0], size=rest)
    rays[5 * k:, :3] = og
    rays[5 * k:, 3:] = unit(dg)
    rays = rays.astype(np.float32)
    want = oracle_ffi.octree_rays(scene, mesh, rays)
    r = Renderer(0)
    r.upload_scene(scene)
    got = r.probe_walk(mesh, rays)
    from relativitypathtracer_amd.renderer import RenderError
    not_a_mesh = int(np.flatnonzero(np.asarray(objs["type"]) != 2)[0])
    for bad in (not_a_mesh, -1, len(objs)):
        with pytest.raises(RenderError):
            r.probe_walk(bad, rays[:4])
    r.close()
    hits = int(want[:, 0].sum())
    assert 0.15 * n < hits < 0.9 * n, hits               # the sample exercises hits and misses
    for w, name in enumerate(("reference layouts", "throughput walk", "latency walk")):
        g = got[:, w, :]
        same = (g.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(want))
        bad = np.flatnonzero(~same.all(axis=1))
        assert bad.size == 0, (name, bad.size, bad[:5], rays[bad[:2]], g[bad[:2]], want[bad[:2]])
