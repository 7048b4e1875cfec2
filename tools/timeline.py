#!/usr/bin/env python3
"""Per-wavefront timeline of one frame (diagnostic kernel variant 11): where is the critical path?"""
import os, sys
# the diagnostic kernels live in the diagnostics build only: make -C relativitypathtracer_amd/csrc diag
os.environ.setdefault("RPT_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "relativitypathtracer_amd", "librpt_hip_diag.so"))
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
# per-wave cycle accounting of the walk: [2] tri-loop cycles, [3] tri iterations, [4] descent cycles, [5] descent
# iterations, [6] rest-of-leaf-step cycles, [7] leaf steps, [8],[9] shader clock at start/end
for i in np.argsort(-dur)[:6]:
    row = t[i]
    tot = max(int(row[9] - row[8]), 1)
    print(f"  wave dur {dur[i]:.1f} us, {tot} cycles: tri {row[2]} cyc / {row[3]} it = {row[2]/max(row[3],1):.0f}; descent {row[4]} cyc / {row[5]} it = {row[4]/max(row[5],1):.0f}; "
          f"leaf-rest {row[6]} cyc / {row[7]} steps = {row[6]/max(row[7],1):.0f}; walk share {(row[2]+row[4]+row[6])/tot:.2f}")
walk = t[:, 2] + t[:, 4] + t[:, 6]
print(f"all waves: tri {t[:,2].sum()/max(t[:,3].sum(),1):.0f} cyc/it ({t[:,3].sum()} it), descent {t[:,4].sum()/max(t[:,5].sum(),1):.0f} cyc/it ({t[:,5].sum()} it), "
      f"leaf-rest {t[:,6].sum()/max(t[:,7].sum(),1):.0f} cyc/step ({t[:,7].sum()} steps); walk cycles {walk.sum():.3g} of {(t[:,9]-t[:,8]).sum():.3g} total wave cycles")
order = np.argsort(-dur)[:8]
print("longest waves (start us, dur us, end us):", [(round(float(start[i]), 1), round(float(dur[i]), 1), round(float(end[i]), 1)) for i in order])
last = np.argsort(-end)[:5]
print("last-finishing waves (start, dur, end):", [(round(float(start[i]), 1), round(float(dur[i]), 1), round(float(end[i]), 1)) for i in last])
# concurrency over time
edges = np.linspace(0, end.max(), 21)
conc = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("waves alive per 5% time slice:", conc)
