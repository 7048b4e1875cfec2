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
