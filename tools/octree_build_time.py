#!/usr/bin/env python3
"""rpt_build_octree on bunny.obj subdivided 1->4 `levels` times (NON-REFERENCE mesh): wall time of the call, repeated.
usage: python tools/octree_build_time.py [levels=2] [repeats=5]      (RPT_OCTREE_TIMING=1 prints the call's phases)"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dense_mesh import subdivide_obj                             # noqa: E402
from relativitypathtracer_amd import Scene                      # noqa: E402
from relativitypathtracer_amd.renderer import Renderer          # noqa: E402
from relativitypathtracer_amd.scene import ASSET_ROOT           # noqa: E402

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 2
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "Models"))
dst = os.path.join(tmp, "Models", "dense.obj")
subdivide_obj(os.path.join(ASSET_ROOT, "Models", "bunny.obj"), dst, levels)
r = Renderer(0)
for k in range(repeats):
    s = Scene(asset_root="/")
    first = s.ReadOBJ(dst, octree=False)
    t0 = time.perf_counter()
    r.build_octree(s, first)
    dt = time.perf_counter() - t0
    b = s.buffers()
    print(f"build {k}: {dt * 1e3:.2f} ms  ({b['triangles'].size // 9} triangles -> {b['octrees'].size // 96} nodes, {b['octreeTris'].size} list entries)", flush=True)
