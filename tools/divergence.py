#!/usr/bin/env python3
"""SIMD utilisation of the octree-walk loops (diagnostic kernel variant 7)."""
import os, sys
# the diagnostic kernels live in the diagnostics build only: make -C relativitypathtracer_amd/csrc diag
os.environ.setdefault("RPT_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "relativitypathtracer_amd", "librpt_hip_diag.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativitypathtracer_amd import Scene
from relativitypathtracer_amd.renderer import Renderer
r = Renderer(0)
for name, v, t in [("bunny", (0, 0, 0), 0.0), ("shadows", (0, 0, 0), 16.0)]:
    s = Scene.from_file(name); s.set_camera(v, t); s.update_objects()
    for W, H in [(1920, 1080), (3840, 2160)]:
        r.set_variant(7); r.upload_scene(s); r.set_scene_params(s, W, H); r.set_output(None); r.render()
        c = r.read_counters()
        names = ["leaf steps", "tri tests", "descent steps"]
        print(f"{name} {W}x{H}: per leaf step: {c[8]/max(c[3],1):.2f} distinct nodes among {c[9]/max(c[3],1):.1f} active lanes; histogram (1,2,3-4,5-8,9-16,>16): {c[10:16]}")
        print(f"{name} {W}x{H}: longest walk {c[6]} leaf steps, walks > 32 steps: {c[7]}")
        print(f"{name} {W}x{H}: " + "; ".join(f"{n}: lanes {c[i]} waves {c[3+i]} util {c[i]/(64*max(c[3+i],1)):.3f}" for i, n in enumerate(names)))
