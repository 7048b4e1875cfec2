#!/usr/bin/env python3
"""Kernel-time ablation on one GPU: where does a frame's time go?

Renders variations of a scene (no objects, light sphere only, mesh only, full) at one resolution and
prints the average render-kernel time of each, measured with HIP events (rpt_timed_frames).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativitypathtracer_amd import Scene                      # noqa: E402
from relativitypathtracer_amd.renderer import Renderer          # noqa: E402

SCENES = {
    "empty": "A0.2\nR\n",
    "sphere_light": "Os\n l1\n p0,2,4,0,0,0,0,0.1,0.1,0.1\n c1,1,1\nA0.2\nR\n",
    "bunny_mesh_only": "MModels/StanfordBunny.obj\nTTextures/StanfordBunnyTerracotta.jpg\nOm0\n p-0.5,-3,5,3.14,0,1,0,20,20,20\n t0\nA0.2\nR\n",
    "bunny": None, "shadows": None, "arch": None, "cube": None, "cubes": None,
}
CAMERA = {"shadows": ((0, 0, 0), 16.0), "arch": ((0, 0, 0.95), 5.25)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--diag", action="store_true", help="librpt_hip_diag.so: the measurement arms (variants other than 0, 1, 3, 41, 43, 44, 50, 51)")
    args = ap.parse_args()
    r = Renderer(0, diag=args.diag)
    r.set_variant(args.variant)
    res = {}
    for name, text in SCENES.items():
        if args.only and name not in args.only.split(","):
            continue
        if text is None:
            s = Scene.from_file(name)
        else:
            s = Scene()
            s.inputScene(text)
        v, t = CAMERA.get(name, ((0, 0, 0), 0.0))
        s.set_camera(v, t)
        s.update_objects()
        r.upload_scene(s)
        r.set_scene_params(s, args.width, args.height)
        r.set_output(None)
        ms = r.timed_frames(args.frames)
        res[name] = round(ms, 4)
        print(f"{name:18s} {ms:8.4f} ms  {args.width*args.height/ms/1e3:10.1f} Mrays/s", flush=True)
    print(json.dumps({"width": args.width, "height": args.height, "variant": args.variant, "kernel_ms": res}))


if __name__ == "__main__":
    main()
