"""GPU octree build (rpt_build_octree) vs the host builder: byte-identical node and
octreeTris arrays — the host builder itself reproduces every count the survey recorded from the reference's."""
import os
import sys
import time

import numpy as np
import pytest

from relativitypathtracer_amd import Scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def renderer():
    from relativitypathtracer_amd.renderer import Renderer
    r = Renderer(0)
    yield r
    r.close()


def build_both(renderer, paths, asset_root=None):
    kw = {} if asset_root is None else {"asset_root": asset_root}
    cpu, gpu = Scene(**kw), Scene(**kw)
    t_cpu = t_gpu = 0.0
    for p in paths:
        t0 = time.perf_counter()
        cpu.ReadOBJ(p)
        t_cpu += time.perf_counter() - t0
        first = gpu.ReadOBJ(p, octree=False)
        t0 = time.perf_counter()
        renderer.build_octree(gpu, first)
        t_gpu += time.perf_counter() - t0
    return cpu, gpu, t_cpu, t_gpu


@pytest.mark.parametrize("paths", [["Models/bunny.obj"], ["Models/pear.obj"], ["Models/cube.obj"], ["Models/triangle.obj"],
                                   ["Models/cube.obj", "Models/pear.obj"], ["Models/pear.obj", "Models/bunny.obj", "Models/triangle.obj"]])
def test_gpu_octree_is_byte_identical(renderer, paths):
    cpu, gpu, _, _ = build_both(renderer, paths)
    a, b = cpu.buffers(), gpu.buffers()
    assert cpu.mesh_roots() == gpu.mesh_roots()
    for k in ("vertices", "normals", "triangles", "octreeTris", "octrees"):
        assert a[k].shape == b[k].shape, k
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), f"{paths}: {k} differs"


def test_gpu_octree_dense_mesh_and_render(renderer, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dense_mesh import subdivide_obj
    from relativitypathtracer_amd.scene import ASSET_ROOT
    os.makedirs(tmp_path / "Models")
    dst = str(tmp_path / "Models" / "bunny_x16.obj")
    subdivide_obj(os.path.join(ASSET_ROOT, "Models", "bunny.obj"), dst, 2)
    cpu, gpu, t_cpu, t_gpu = build_both(renderer, [dst], asset_root="/")
    a, b = cpu.buffers(), gpu.buffers()
    assert a["octrees"].size // 96 == 38361 and a["octreeTris"].size == 790266
    assert np.array_equal(a["octrees"], b["octrees"]) and np.array_equal(a["octreeTris"], b["octreeTris"])
    print(f"octree build, 79 488 triangles: host {t_cpu*1e3:.0f} ms (incl. OBJ parse), device path {t_gpu*1e3:.0f} ms")


def test_gpu_octree_repeats_the_build_when_a_list_outgrows_its_buffer(renderer, monkeypatch):
    """The build is one submission with list buffers of a fixed capacity; a level that outgrows it raises a flag in the
    device-side header (nothing is written out of bounds) and the host repeats the build with four times the capacity.
    Started at the smallest capacity possible (the root's own list), bunny.obj needs two repeats."""
    monkeypatch.setenv("RPT_OCTREE_LIST_CAP", "1")
    cpu, gpu, _, _ = build_both(renderer, ["Models/bunny.obj"])
    a, b = cpu.buffers(), gpu.buffers()
    assert np.array_equal(a["octrees"], b["octrees"]) and np.array_equal(a["octreeTris"], b["octreeTris"])


@pytest.mark.parametrize("seed", range(16))
def test_gpu_octree_random_triangle_soups(renderer, tmp_path, seed):
    """Random meshes the shipped ones do not resemble: triangle soups of 1 .. 3000 triangles — tiny, huge and sliver triangles,
    zero-area ones, vertices shared by most of the mesh (the valence stop rule), all triangles in one corner of the box, flat
    meshes (a root box of zero thickness) — built by the host builder and by the one-submission device builder: byte-identical."""
    rng = np.random.default_rng(4242 + seed)
    n_v = int(rng.choice([3, 8, 50, 400, 1500]))
    n_t = int(rng.choice([1, 5, 60, 700, 3000]))
    scale = rng.choice([1e-3, 1.0, 1.0, 50.0])
    verts = rng.normal(size=(n_v, 3)) * scale
    mode = seed % 4
    if mode == 1:
        verts[:, 2] = 0.25                                   # a flat mesh
    if mode == 2:
        verts[: n_v // 2] = verts[0] + rng.normal(size=(n_v // 2, 3)) * 1e-4 * scale      # half of the vertices in one spot
    tris = rng.integers(0, n_v, size=(n_t, 3))
    if mode == 3:
        tris[:, 0] = 0                                        # one vertex in every triangle
    degenerate = rng.random(n_t) < 0.05
    tris[degenerate, 2] = tris[degenerate, 1]               # a few zero-area triangles
    os.makedirs(tmp_path / "Models")
    path = str(tmp_path / "Models" / f"soup{seed}.obj")
    with open(path, "w") as f:
        for v in verts:
            f.write(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for t in tris:
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
    cpu, gpu, _, _ = build_both(renderer, [path], asset_root="/")
    a, b = cpu.buffers(), gpu.buffers()
    for k in ("vertices", "normals", "triangles", "octreeTris", "octrees"):
        assert a[k].shape == b[k].shape, (seed, k, a[k].shape, b[k].shape)
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), f"soup {seed}: {k} differs"
