#!/usr/bin/env python3
"""Where do a frame's vector instructions go?  Renders one scene in a few reduced forms, one launch each, so that
`rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES` lists them in order:
    0 the frame as it is (default kernel)        1 light propagation off (interval 0: no lights, no shadow rays)
    2 every object tested for every pixel (variant 3, no culling)
usage: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d out -o x -- python3 tools/valu_breakdown.py shadows 3840 2160"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativitypathtracer_amd import Scene                      # noqa: E402
from relativitypathtracer_amd.renderer import Renderer          # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "shadows"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
cam = {"shadows": ((0, 0, 0), 16.0), "arch": ((0, 0, 0.95), 5.25)}.get(name, ((0, 0, 0), 0.0))
s = Scene.from_file(name)
s.set_camera(*cam)
s.update_objects()
r = Renderer(0)
r.upload_scene(s)
r.set_scene_params(s, W, H)
r.set_output(None)
r.set_variant(41)
r.render_async(); r.sync()                 # 0
s.set_interval(0); s.update_objects()
r.set_scene_params(s, W, H); r.set_objects(s)
r.render_async(); r.sync()                 # 1
s.set_interval(-1); s.update_objects()
r.set_scene_params(s, W, H); r.set_objects(s)
r.set_variant(3)
r.render_async(); r.sync()                 # 2
r.close()
