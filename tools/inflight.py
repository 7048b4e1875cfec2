#!/usr/bin/env python3
"""Render-stage throughput of ONE rank with several frames in flight (one GPU).

Emulates the render stage of rank 0 in an N-rank job (tiles 0, N, 2N, ... into a colour plane) with M frames in
flight: M contexts, each on its own stream, each rendering `frames` frames back to back (rpt_set_objects +
rpt_render_async).  Prints the wall time per frame of the whole set, i.e. what bounds the frame rate of a
pipelined multi-GPU job before the exchange step.  The exchange itself needs more than one GPU and is not here.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from relativitypathtracer_amd import Scene                      # noqa: E402
from relativitypathtracer_amd.renderer import Renderer          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--world", default="1,2,4,8")
    ap.add_argument("--inflight", default="1,2,3,4")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-objects", action="store_true", help="do not refresh Object[] every frame (isolates the copy)")
    args = ap.parse_args()
    import ablate
    if args.scene == "dense":                       # NON-REFERENCE: bunny.obj subdivided to 79 488 triangles (tools/dense_mesh.py)
        import dense_mesh
        s = dense_mesh.dense_bunny_scene("/tmp/rpt_dense", 2)
    elif ablate.SCENES.get(args.scene):               # the inline ablation scenes of tools/ablate.py (empty, sphere_light, ...)
        s = Scene()
        s.inputScene(ablate.SCENES[args.scene])
    else:
        s = Scene.from_file(args.scene)
    s.set_camera((0, 0, 0), 16.0 if args.scene == "shadows" else 0.0)
    s.update_objects()
    mmax = max(int(m) for m in args.inflight.split(","))
    ctxs = []
    for _ in range(mmax):
        r = Renderer(0)
        r.set_variant(args.variant)
        if ctxs:
            r.share_scene(ctxs[0])
        else:
            r.upload_scene(s)
        r.set_scene_params(s, args.width, args.height)
        ctxs.append(r)
    for world in [int(w) for w in args.world.split(",")]:
        for r in ctxs:
            if world == 1:
                r.set_rows(0, 1, False)
                r.set_output(None)
            else:
                r.set_rows(0, world, True)
                r.set_plane_output(None)
        for m in [int(m) for m in args.inflight.split(",")]:
            use = ctxs[:m]
            for _ in range(5):
                for r in use:
                    r.set_objects(s)
                    r.render_async()
            for r in use:
                r.sync()
            t0 = time.perf_counter()
            for _ in range(args.frames):
                for r in use:
                    if not args.no_objects:
                        r.set_objects(s)
                    r.render_async()
            for r in use:
                r.sync()
            dt = (time.perf_counter() - t0) * 1e3 / (args.frames * m)
            print(f"world {world}  in flight {m}:  {dt:7.4f} ms/frame per rank  "
                  f"-> job {args.width * args.height / dt / 1e3:9.0f} Mrays/s if the exchange keeps up", flush=True)


if __name__ == "__main__":
    main()
