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


# ---- object-level known-answer tests (intersect_sphere, intersect_cube, transformPoint*, sample_light) -----------

KAT_SCENE = """MModels/bunny.obj
MModels/pear.obj
TTextures/soccer.jpg
Os
 p-5,-3,10,0,0,0,0,0.5,0.5,0.5
 l1
 c10,10,10
Os
 p-1,0.5,7,0.7,0.3,1,0.2,1.5,0.4,0.9
 c0.9,0.1,0.2
 t0
 v0.3,0.1,-0.5
Oc
 p0,-4,5,0,0,0,0,10,0.5,10
 c1,1,1
Oc
 p3,1,9,1.1,1,1,1,0.6,2,0.3
 c0.2,0.9,0.2
 v-0.6,0,0.2
Om1
 p2,-1.5,12,0.4,0,1,0,0.5,0.5,0.5
 c0.85,0.86,0.40
Om0
 p-3,-2,9,3.14,0,1,0,12,12,12
 c0.8,0.5,0.3
 v0,0.2,0
A0.2
R
"""


def _kat_scene(v=(0.2, -0.1, 0.4), t=3.0):
    from relativitypathtracer_amd import Scene
    s = Scene()
    s.inputScene(KAT_SCENE)
    s.set_camera(v, t)
    s.update_objects()
    return s


def _same(got, want):
    return ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)))


def _rest_frame_rays(scene, i, rng, n):
    """4-D rays in object i's REST frame that exercise its intersector: origins outside, inside, on faces; directions at the shape,
    tangent to it, along its faces, away from it.  Built in object space and mapped with M (so that InvM brings them back)."""
    o = scene.objects()[i]
    M = np.array(o["M"], dtype=np.float64).reshape(4, 4)
    kind = int(o["type"])
    if kind == 2:
        node = scene.octrees()[int(o["meshIndex"])]
        lo, hi = np.array(node["min"][:3], dtype=np.float64), np.array(node["max"][:3], dtype=np.float64)
    else:
        lo, hi = -np.ones(3), np.ones(3)
    c, h = 0.5 * (lo + hi), 0.5 * (hi - lo)
    unit = lambda v: v / np.linalg.norm(v, axis=-1, keepdims=True)      # noqa: E731
    org = c + unit(rng.normal(size=(n, 3))) * h * np.exp(rng.uniform(np.log(1.2), np.log(40.0), size=(n, 1)))
    inside = rng.random(n) < 0.2
    org[inside] = c + rng.uniform(-1, 1, size=(int(inside.sum()), 3)) * h * 0.95           # inside the cube / sphere / root box (winding -1)
    onface = rng.random(n) < 0.1
    ax = rng.integers(0, 3, size=n)
    org[onface, ax[onface]] = (c + h * rng.choice([-1.0, 1.0], size=(n, 1)))[onface, ax[onface]]   # exactly on a face plane
    target = c + rng.normal(size=(n, 3)) * h * rng.choice([0.3, 1.0, 1.0, 1.5], size=(n, 1))
    graz = rng.random(n) < 0.2
    target[graz] = (c + unit(rng.normal(size=(n, 3))) * h * (1.0 + rng.choice([-1e-6, 0.0, 1e-6, 1e-3], size=(n, 1))))[graz]   # sphere tangents / cube edges
    d = target - org
    along = rng.random(n) < 0.1
    d[along, ax[along]] = 0.0                                                               # parallel to a pair of faces
    d[rng.random(n) < 0.1] *= -1.0                                                          # pointing away
    d = d * np.exp(rng.uniform(-2, 2, size=(n, 1)))
    wo = org @ M[:3, :3].T + M[:3, 3]
    wd = d @ M[:3, :3].T
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0] = rng.uniform(-5, 5, size=n)
    rays[:, 1:4] = wo
    rays[:, 4] = rng.choice([-1.0, 0.0, -2.5], size=n)
    rays[:, 5:8] = wd
    return rays


@pytest.fixture(scope="module")
def kat():
    from relativitypathtracer_amd.renderer import Renderer
    scene = _kat_scene()
    r = Renderer(0)
    r.upload_scene(scene)
    r.set_scene_params(scene, 64, 64)
    yield scene, r
    r.close()


@pytest.mark.parametrize("obj", [0, 1, 2, 3, 4, 5])
def test_intersectors_at_ray_level(kat, obj):
    """intersect_sphere (opencl_kernel.cl:335-359), intersect_cube (:312-333, winding -1 for inside origins included) and
    intersect_octree (:200-308) in the general 4-D form, ray by ray against the oracle: hit flag, distance, normal, (u, v)."""
    scene, r = kat
    rng = np.random.default_rng(100 + obj)
    rays = _rest_frame_rays(scene, obj, rng, 20000)
    want = oracle_ffi.object_rays(scene, 0, obj, rays)
    got = r.probe_object(0, obj, rays)
    hits = int(want[:, 0].sum())
    assert 0.1 * len(rays) < hits < 0.95 * len(rays), hits
    o = scene.objects()[obj]
    if int(o["type"]) == 0 and int(o["textureIndex"]) == -1:
        want[:, 5:7] = 0.0          # the (u, v) of a sphere hit is only ever read by the texture fetch: not evaluated for untextured spheres
    bad = np.flatnonzero(~_same(got, want).all(axis=1))
    assert bad.size == 0, (obj, bad.size, rays[bad[:2]], got[bad[:2]], want[bad[:2]])


def test_transforms_match_the_oracle(kat):
    """transformPoint, transformPoint4D, transformDirection, applyTranspose (opencl_kernel.cl:75-104) on random and extreme vectors."""
    scene, r = kat
    rng = np.random.default_rng(5)
    v = rng.normal(size=(8192, 4)) * np.exp(rng.uniform(-20, 20, size=(8192, 1)))
    v[:8] = [[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [-0.0, 0.0, -0.0, 0.0], [3e38, -3e38, 1e-38, 1e-45], [np.inf, 1, 1, 1]]
    v = v.astype(np.float32)
    for obj in range(6):
        want = oracle_ffi.object_rays(scene, 2, obj, v)
        got = r.probe_object(2, obj, v)
        assert _same(got, want).all(), obj


def test_primary_ray_form_equals_the_general_form(kat):
    """The default kernels' primary rays take the origin and its constants from the per-frame DObj record and form only rows 1..3 of
    Lorentz * (interval, nd) (intersect_object_primary); the reference feeds stationaryCam and the full product to the same
    intersectors (opencl_kernel.cl:382-390).  Same hits, bit for bit, for every object type."""
    scene, r = kat
    rng = np.random.default_rng(9)
    n = 30000
    cam = rng.normal(size=(n, 3)) * [1.0, 0.6, 0.3] + [0, 0, 0.5]
    cam = cam.astype(np.float32)
    nd = cam / np.sqrt(((cam[:, 0] * cam[:, 0] + cam[:, 1] * cam[:, 1]) + cam[:, 2] * cam[:, 2]).astype(np.float32))[:, None]
    nd = nd.astype(np.float32)
    interval = scene.params["interval"]
    any_hits = 0
    for obj in range(6):
        o = scene.objects()[obj]
        l4 = np.concatenate([np.full((n, 1), interval, np.float32), nd], axis=1)
        # transformPoint4D(Lorentz, (interval, nd)): the oracle's transform entry takes {x, y, z, w} = the 4-vector's four components in order
        boosted = oracle_ffi.object_rays(scene, 2, obj, l4)[:, 4:8]
        rays = np.concatenate([np.tile(np.asarray(o["stationaryCam"], dtype=np.float32), (n, 1)), boosted], axis=1)
        want = oracle_ffi.object_rays(scene, 0, obj, rays)
        got = r.probe_object(3, obj, cam)
        any_hits += int(want[:, 0].sum())
        if int(o["type"]) == 0 and int(o["textureIndex"]) == -1:
            want[:, 5:7] = 0.0
        bad = np.flatnonzero(~_same(got, want).all(axis=1))
        assert bad.size == 0, (obj, bad.size, cam[bad[:2]], got[bad[:2]], want[bad[:2]])
    assert any_hits > 1000


def _shadow_rays(scene, rng, n, light):
    """Shadow rays as trace() makes them (opencl_kernel.cl:574-596): from points on and around the objects towards the light, in waves of
    64 lanes that share their region (so that whole waves can be culled), light distances from 'right at the origin' to 'far behind
    every object'."""
    objs = scene.objects()
    Ml = np.array(objs[light]["M"], dtype=np.float64).reshape(4, 4)
    lpos = Ml[:3, 3]
    rays = np.zeros((n, 9), dtype=np.float32)
    for w0 in range(0, n, 64):
        k = min(64, n - w0)
        j = int(rng.integers(0, len(objs)))
        M = np.array(objs[j]["M"], dtype=np.float64).reshape(4, 4)
        centre = M[:3, 3] + rng.normal(size=3) * rng.choice([0.0, 0.5, 3.0, 20.0])
        p = centre + rng.normal(size=(k, 3)) * rng.choice([0.01, 0.3, 2.0])
        target = lpos + rng.normal(size=3) * rng.choice([0.0, 0.0, 1.0, 10.0])
        d = target - p
        dist = np.linalg.norm(d, axis=1)
        rays[w0:w0 + k, 0] = rng.uniform(-3, 3)
        rays[w0:w0 + k, 1:4] = p
        rays[w0:w0 + k, 4] = -dist                                   # (time component as trace() sets it: minus the distance)
        rays[w0:w0 + k, 5:8] = d
        rays[w0:w0 + k, 8] = dist * rng.choice([1.0, 1.0, 0.5, 0.05, 2.0, 30.0])
    return rays


@pytest.mark.parametrize("state", [((0.0, 0.0, 0.0), 0.0), ((0.2, -0.1, 0.4), 3.0), ((0.0, 0.0, 0.95), 7.0)])
def test_sample_light_at_ray_level(state):
    """sample_light (opencl_kernel.cl:488-545) on shadow rays the frames do and do not contain: the un-culled form equals the oracle's
    first-occluder answer, and the culled form — unit_segment_apart, mesh_ray_misses_root, mesh_segment_apart over whole waves —
    equals the un-culled one.  The bunny in the scene keeps its segment cull (small triangles), the pear only the ray test."""
    from relativitypathtracer_amd.renderer import Renderer
    scene = _kat_scene(*state)
    r = Renderer(0)
    r.upload_scene(scene)
    r.set_scene_params(scene, 64, 64)
    rec_pear, rec_bunny = r.mesh_segment_cull_record(4), r.mesh_segment_cull_record(5)
    # the pear: triangles up to half a unit long in a model five units tall, 16.2 K = 3.1 — no margin short of the mesh's own size
    # makes a segment cull provable, it keeps the ray test only.  (As the scene's SECOND mesh its lists also carry the bunny's
    # triangles, Mesh.cpp:16-19 — which happen to lie inside the pear's root box.)
    assert rec_pear[0] > 0 and rec_pear[4] < 0 and rec_pear[7] > 0.05, rec_pear
    assert rec_bunny[0] > 0 and 0 < rec_bunny[4] < 0.01 and rec_bunny[9] == 1.0 and 1e-4 <= rec_bunny[6] < 1e-2, rec_bunny
    rng = np.random.default_rng(77)
    rays = _shadow_rays(scene, rng, 64 * 3000, 0)
    want = oracle_ffi.object_rays(scene, 1, 0, rays) >= 0
    got = r.probe_object(1, 0, rays) > 0
    r.close()
    assert 0.05 < want.mean() < 0.95, want.mean()
    bad = np.flatnonzero(got[:, 0] != want)
    assert bad.size == 0, ("un-culled sample_light differs from the oracle", bad.size, rays[bad[:2]])
    bad = np.flatnonzero(got[:, 1] != want)
    assert bad.size == 0, ("the segment culls changed an answer", bad.size, rays[bad[:2]])
