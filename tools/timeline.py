#!/usr/bin/env python3
"""Per-wavefront timeline of one frame (diagnostic kernel variant 7): where is the critical path?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativitypathtracer_amd import Scene
from relativitypathtracer_amd.renderer import Renderer
name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
cam = {"shadows": ((0, 0, 0), 16.0), "arch": ((0, 0, 0.95), 5.25)}.get(name, ((0, 0, 0), 0.0))
s = Scene.from_file(name); s.set_camera(*cam); s.update_objects()
r = Renderer(0); r.set_variant(11); r.upload_scene(s); r.set_scene_params(s, W, H); r.set_output(None)
r.render(); r.render()
t = r.read_wave_times().astype(np.int64)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0      # microseconds
dur = end - start
print(f"{name} {W}x{H}: waves {len(t)}, kernel span {end.max():.1f} us (diag build), sum of wave durations {dur.sum()/1e3:.1f} ms")
print("duration percentiles us:", {p: round(float(np.percentile(dur, p)), 2) for p in (50, 90, 99, 99.9, 100)})
order = np.argsort(-dur)[:8]
print("longest waves (start us, dur us, end us):", [(round(float(start[i]), 1), round(float(dur[i]), 1), round(float(end[i]), 1)) for i in order])
last = np.argsort(-end)[:5]
print("last-finishing waves (start, dur, end):", [(round(float(start[i]), 1), round(float(dur[i]), 1), round(float(end[i]), 1)) for i in last])
# concurrency over time
edges = np.linspace(0, end.max(), 21)
conc = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("waves alive per 5% time slice:", conc)
