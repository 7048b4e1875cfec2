#!/usr/bin/env python3
"""What do the screen bounds cost the host per object — proposal (outline sampling) and proof (rpt_bounds_certify.hpp)?  Pure host
code: the objects of the shipped scenes along a camera sweep (rest -> 0.99c, clock 0 -> 30 s) and of the scene generators are
written to a file and timed by tools/native/bounds_cost.cpp (g++ -O2, one thread, best of 7 passes).
usage: python tools/bounds_cost.py > profiles/r04_bounds_host_cost.txt"""
import math
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np                                               # noqa: E402
from relativitypathtracer_amd import Scene                       # noqa: E402
import verify_fuzz                                               # noqa: E402


def records(scene):
    objs, nodes = scene.objects(), scene.octrees()
    out = b""
    for i in range(min(len(objs), 64)):
        root = np.zeros(6, dtype=np.float32)
        has_root = int(objs["type"][i]) == 2
        if has_root:
            n = nodes[int(objs["meshIndex"][i])]
            root[:3], root[3:] = n["min"][:3], n["max"][:3]
        out += struct.pack("ii", scene.params["interval"], int(has_root)) + root.tobytes() + objs[i:i + 1].tobytes()
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "bounds_cost")
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "native", "bounds_cost.cpp")], check=True)
        cpu = ""
        try:
            cpu = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
        except Exception:
            pass
        print(f"host cost of the screen bounds per object, one thread of: {cpu}")
        groups = []
        for name in ("cube", "arch", "bunny", "shadows", "cubes", "soccer", "rulers", "ladder_paradox"):
            s = Scene.from_file(name)
            still = b""
            s.update_objects()
            still = records(s)
            groups.append((f"{name} (camera at rest, t = 0)", still))
            blob = b""
            for k in range(100):
                f = k / 99
                speed = 0.99 * f
                ang, el = 2.0 * math.pi * 3.0 * f, 0.6 * math.sin(2.0 * math.pi * 5.0 * f)
                s.set_camera((speed * math.cos(el) * math.sin(ang), speed * math.sin(el), speed * math.cos(el) * math.cos(ang)), 30.0 * f)
                s.update_objects()
                blob += records(s)
            groups.append((f"{name} (sweep: rest -> 0.99c, 100 states)", blob))
        for kind in ("random", "extreme", "close", "walls", "ellipsoids", "meshwalls"):
            blob = b""
            for seed in range(200):
                try:
                    s, _ = verify_fuzz.build(kind, seed)
                except Exception:
                    continue
                blob += records(s)
            groups.append((f"generator {kind} (200 scenes)", blob))
        for label, blob in groups:
            path = os.path.join(tmp, "objects.bin")
            with open(path, "wb") as f:
                f.write(blob)
            out = subprocess.run([exe, path], capture_output=True, text=True).stdout.strip()
            print(f"{label:44s} {out}", flush=True)


if __name__ == "__main__":
    main()
