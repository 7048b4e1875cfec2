#!/usr/bin/env python3
"""Every BASELINE.json configuration (+ the reference's other scenes) for a list of kernel variants: ms/frame one frame at a
time (submit, wait) and with three frames in flight, through the same C-ABI calls bench.py makes per frame
(rpt_set_objects + rpt_render_async).  usage: python tools/configs.py --variants 26,41 [--frames 40]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativitypathtracer_amd import Scene                      # noqa: E402
from relativitypathtracer_amd.renderer import Renderer          # noqa: E402

CONFIGS = [("cube", 640, 480, (0, 0, 0), 0.0), ("arch", 1920, 1080, (0, 0, 0.95), 5.25), ("bunny", 1920, 1080, (0, 0, 0), 0.0),
           ("shadows", 3840, 2160, (0, 0, 0), 16.0), ("bunny", 3840, 2160, (0, 0, 0), 0.0), ("bunny", 7680, 4320, (0, 0, 0), 0.0),
           ("arch", 3840, 2160, (0, 0, 0.95), 5.25), ("cubes", 3840, 2160, (0.3, 0, 0.1), 3.0), ("rulers", 3840, 2160, (0, 0, 0), 2.5),
           ("ladder_paradox", 3840, 2160, (0, 0, 0), 1.0), ("soccer", 3840, 2160, (0, 0, 0), 2.0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--sizes", default="", help="WxH,WxH,...: render each selected scene (first entry of its name) at these sizes instead")
    ap.add_argument("--diag", action="store_true", help="librpt_hip_diag.so: the measurement arms (variants other than 0, 1, 3, 41, 43, 44, 50, 51)")
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    rows = []
    configs = CONFIGS
    if args.sizes:
        first = {}
        for c in CONFIGS:
            first.setdefault(c[0], c)
        configs = [(n, int(wh.split("x")[0]), int(wh.split("x")[1]), first[n][3], first[n][4])
                   for n in (args.only.split(",") if args.only else first) for wh in args.sizes.split(",")]
    for name, W, H, vel, t in configs:
        if args.only and name not in args.only.split(","):
            continue
        s = Scene.from_file(name)
        s.set_camera(vel, t)
        s.update_objects()
        slots = [Renderer(0, diag=args.diag) for _ in range(args.inflight)]
        slots[0].upload_scene(s)
        for r in slots[1:]:
            r.share_scene(slots[0])
        for r in slots:
            r.set_scene_params(s, W, H)
            r.set_output(None)
        for v in variants:
            for r in slots:
                r.set_variant(v)
                r.set_objects(s)
                r.render()
            t0 = time.perf_counter()
            for _ in range(args.frames):
                slots[0].set_objects(s)
                slots[0].render()
            one = (time.perf_counter() - t0) / args.frames * 1e3
            t0 = time.perf_counter()
            for f in range(args.frames * 3):
                r = slots[f % len(slots)]
                r.sync()
                r.set_objects(s)
                r.render_async()
            for r in slots:
                r.sync()
            fl = (time.perf_counter() - t0) / (args.frames * 3) * 1e3
            rows.append({"scene": name, "W": W, "H": H, "variant": v, "ms_one_at_a_time": round(one, 4), "ms_in_flight": round(fl, 4),
                         "mrays_in_flight": round(W * H / fl / 1e3, 1)})
            print(f"{name:16s} {W}x{H} variant {v:3d}: one at a time {one:8.4f} ms   {args.inflight} in flight {fl:8.4f} ms/frame "
                  f"({W * H / fl / 1e3:9.1f} Mrays/s)", flush=True)
        for r in slots:
            r.close()
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
