import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_ffi
from relativitypathtracer_amd import Scene
from relativitypathtracer_amd.renderer import Renderer
name, k, states = sys.argv[1], int(sys.argv[2]), 1000
W, H = 640, 360
s = Scene.from_file(name)
f = k / (states - 1)
speed = 0.99 * f
ang, el = 2.0 * math.pi * 3.0 * f, 0.6 * math.sin(2.0 * math.pi * 5.0 * f)
v = (speed * math.cos(el) * math.sin(ang), speed * math.sin(el), speed * math.cos(el) * math.cos(ang))
s.set_camera(v, 30.0 * f); s.update_objects()
opx, orgb, _ = oracle_ffi.render(s, W, H)
r = Renderer(0)
r.upload_scene(s); r.set_scene_params(s, W, H); r.set_output(None)
fr = {}
for variant in (41, 43, 3, 1):
    r.set_variant(variant); r.set_objects(s); r.render()
    px = r.read_framebuffer()
    fr[variant] = px["rgba"].copy()
    print(variant, int((px["rgba"] != opx["rgba"]).any(axis=1).sum()), "pixels differ from the oracle")
print("41 vs 3:", int((fr[41] != fr[3]).any(axis=1).sum()))
for variant in (0, 43, 3):
    r.set_variant(variant); print("verify", variant, r.verify_frame(), r.verify_frame())
