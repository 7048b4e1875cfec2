#!/usr/bin/env python3
"""rpt_verify_frame over an animated sweep of every shipped scene: the camera goes from rest to 0.99c along a direction that
turns as it accelerates while the clock runs from 0 to 30 s, `--states` states per scene; every state is checked on the device —
the kernel a frame gets (41, the asynchronous default, and 43, the blocking one) against the un-culled kernel (3), packed colours
of every pixel.  A lost pixel (an object culled from a tile it is visible in, a shadow ray culled that was occluded) would be
silent in production; here it is a number.  usage: python tools/verify_sweep.py [--states 1000] [--width 640 --height 360]"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relativitypathtracer_amd import Scene                      # noqa: E402
from relativitypathtracer_amd.renderer import Renderer          # noqa: E402

SCENES = ["cube", "arch", "bunny", "shadows", "cubes", "rulers", "ladder_paradox", "soccer"]


# Explicit kernels: since the small-frame rule (rpt_api.hip: frames of at most RPT_LATENCY_KERNEL_MAX_PIXELS get 43 from the
# asynchronous call too) variant 0 resolves to 43 at every size a soak uses, so "0 and 43" ran kernel 43 twice (round 3's records
# did).  41 = the throughput kernel of the 4K headline, 43 = the latency kernel, 0 = whatever a frame would get (44 on
# scenes without meshes).
VERIFIED_VARIANTS = (41, 43, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--states", type=int, default=1000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--path", type=int, default=0, help="0: the path of the docstring; k > 0: other turning rates and phases (a different walk through the same states)")
    args = ap.parse_args()
    r = Renderer(0)
    total_bad = 0
    for name in SCENES:
        s = Scene.from_file(name)
        s.update_objects()
        r.upload_scene(s)
        r.set_scene_params(s, args.width, args.height)
        r.set_output(None)
        bad, worst, t0 = 0, 0, time.perf_counter()
        for k in range(args.states):
            f = k / max(args.states - 1, 1)
            speed = 0.99 * f
            ang, el = 2.0 * math.pi * (3.0 + 1.37 * args.path) * f + 0.9 * args.path, (0.6 + 0.13 * (args.path % 5)) * math.sin(2.0 * math.pi * (5.0 + 0.71 * args.path) * f + 0.4 * args.path)
            v = (speed * math.cos(el) * math.sin(ang), speed * math.sin(el), speed * math.cos(el) * math.cos(ang))
            s.set_camera(v, 30.0 * f)
            s.update_objects()
            r.set_objects(s)
            for variant in VERIFIED_VARIANTS:
                r.set_variant(variant)
                n = r.verify_frame()
                if n:
                    bad += 1
                    worst = max(worst, n)
                    print(f"  {name}: state {k} (v = {v}, t = {30.0 * f:.3f}) kernel {r.last_variant()} (variant {variant}): {n} pixels differ", flush=True)
        dt = time.perf_counter() - t0
        total_bad += bad
        print(f"{name:16s} {args.states} states x 3 selections (41, 43, default) at {args.width}x{args.height}: {bad} states with differences (worst {worst} px), {dt:.1f} s", flush=True)
    r.close()
    print(f"TOTAL: {total_bad} states with differences")
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
