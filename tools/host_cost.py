#!/usr/bin/env python3
"""Host time of one animated frame, call by call (scene update, rpt_set_objects, rpt_render_async), three frames in flight:
is an animated many-object scene bound by the host or by the device?   usage: python tools/host_cost.py [scene] [W H]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativitypathtracer_amd import Scene                      # noqa: E402
from relativitypathtracer_amd.renderer import Renderer          # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cubes"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
cam = {"cubes": ((0.3, 0, 0.1), 3.0), "shadows": ((0, 0, 0), 16.0), "arch": ((0, 0, 0.95), 5.25)}.get(name, ((0, 0, 0), 0.0))
if "RPT_T0" in os.environ:
    cam = (cam[0], float(os.environ["RPT_T0"]))
s = Scene.from_file(name)
s.set_camera(*cam)
s.update_objects()
slots = [Renderer(0, diag=os.environ.get("RPT_HOST_PROFILE") is not None) for _ in range(3)]
slots[0].upload_scene(s)
for r in slots[1:]:
    r.share_scene(slots[0])
for r in slots:
    r.set_scene_params(s, W, H)
    r.set_output(None)
for animate in (False, True):
    for r in slots:
        r.set_objects(s); r.render_async()
    for r in slots:
        r.sync()
    t_upd = t_set = t_ren = 0.0
    n = 300
    clock = cam[1]
    t0 = time.perf_counter()
    for f in range(n):
        a = time.perf_counter()
        if animate:
            clock += 0.016
            s.set_camera(cam[0], clock)
            s.update_objects()
        b = time.perf_counter()
        slots[f % 3].set_objects(s)
        c = time.perf_counter()
        slots[f % 3].render_async()
        d = time.perf_counter()
        t_upd += b - a; t_set += c - b; t_ren += d - c
    submit = time.perf_counter() - t0
    for r in slots:
        r.sync()
    total = time.perf_counter() - t0
    print(f"{name} {W}x{H} t0={cam[1]} {'animated' if animate else 'still'}: {total / n * 1e3:.4f} ms/frame; host per frame: update {t_upd / n * 1e6:.1f} us, "
          f"rpt_set_objects {t_set / n * 1e6:.1f} us, rpt_render_async {t_ren / n * 1e6:.1f} us, all submitted after {submit / n * 1e3:.4f} ms/frame")

# the same calls with the device idle before each (no back-pressure from frames in flight): the host's own cost, the kernel's own time
import ctypes as C
lib = slots[0]._lib
lib.rpt_object_screen_bounds.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
out = (C.c_float * 8)()
t_set = t_rect = t_gpu = 0.0
n = 200
clock = cam[1]
for f in range(n):
    clock += 0.016
    s.set_camera(cam[0], clock)
    s.update_objects()
    r = slots[0]
    b = time.perf_counter()
    r.set_objects(s)
    c = time.perf_counter()
    r.render()
    t_gpu += r.last_frame_ms()
    objs = s.objects()
    base = objs.ctypes.data
    e = time.perf_counter()
    for i in range(len(objs)):
        lib.rpt_object_screen_bounds(base + 320 * i, s.params["interval"], None, out)
    g = time.perf_counter()
    t_set += c - b; t_rect += g - e
print(f"{name} animated, device idle before every call: rpt_set_objects {t_set / n * 1e6:.1f} us of which screen bounds <= {t_rect / n * 1e6:.1f} us "
      f"({len(objs)} objects, through ctypes); kernel {t_gpu / n:.4f} ms/frame one at a time")
for r in slots:
    r.close()          # (RPT_HOST_PROFILE=1, diagnostics library: rpt_destroy prints where rpt_set_objects spent its time)
