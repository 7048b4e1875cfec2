#!/usr/bin/env python3
"""How often does the proof of the screen bounds succeed, and what does it cost?  (Host code only: no GPU.)
For every object of every shipped scene at a sweep of camera states, and of the scene generators of tests/scene_fuzz.py:
the proposal of csrc/rpt_screen_bounds.hpp (rpt_object_screen_bounds_proposed) handed to csrc/rpt_bounds_certify.hpp
(rpt_certify_screen_bounds).  Prints, per group: objects, proposals that claim something, of those proven, the reasons of the
rest, segment tests per proven object (mean / max) and microseconds per object for proposal and proof.
usage: python tools/certify_stats.py [--seeds 300] [--states 200]"""
import argparse
import collections
import ctypes as C
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np                                              # noqa: E402
from relativitypathtracer_amd import Scene, _ffi                 # noqa: E402

REASONS = {0: "proven", 1: "non-finite", 2: "not homeomorphic", 3: "float noise", 4: "origin near shape", 5: "witness", 6: "budget", 7: "hit on boundary"}
FULL = 3.0e38


def objects_of(scene):
    objs, nodes = scene.objects(), scene.octrees()
    for i in range(min(len(objs), 64)):
        raw = objs[i:i + 1].copy()
        root = None
        if int(objs["type"][i]) == 2:
            n = nodes[int(objs["meshIndex"][i])]
            root = (C.c_float * 6)(*n["min"][:3], *n["max"][:3])
        yield raw, root


class Tally:
    def __init__(self):
        self.n = self.claims = self.proven = 0
        self.reasons = collections.Counter()
        self.tests = []
        self.t_prop = self.t_cert = 0.0

    def add(self, lib, raw, root, interval):
        b = (C.c_float * 8)()
        st = (C.c_int * 4)()
        t0 = time.perf_counter()
        lib.rpt_object_screen_bounds_proposed(raw.ctypes.data, interval, root, b)
        t1 = time.perf_counter()
        self.n += 1
        self.t_prop += t1 - t0
        if b[0] <= -FULL and b[1] <= -FULL and b[2] >= FULL and b[3] >= FULL and b[4] <= -FULL and b[5] >= FULL and b[6] <= -FULL and b[7] >= FULL:
            return
        self.claims += 1
        t1 = time.perf_counter()
        ok = lib.rpt_certify_screen_bounds(raw.ctypes.data, interval, root, b, st)
        self.t_cert += time.perf_counter() - t1
        self.reasons[REASONS[st[0]]] += 1
        if ok:
            self.proven += 1
            self.tests.append(st[1])

    def line(self, name):
        rest = ", ".join(f"{k} {v}" for k, v in sorted(self.reasons.items()) if k != "proven")
        tm = (sum(self.tests) / len(self.tests), max(self.tests)) if self.tests else (0, 0)
        return (f"{name:22s} objects {self.n:7d}  claims {self.claims:7d}  proven {self.proven:7d} ({100.0 * self.proven / max(self.claims, 1):6.2f} %)  "
                f"tests/proven {tm[0]:6.1f} max {tm[1]:4d}  proposal {1e6 * self.t_prop / max(self.n, 1):6.2f} us  proof {1e6 * self.t_cert / max(self.claims, 1):6.2f} us  [{rest}]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=300)
    ap.add_argument("--states", type=int, default=200)
    ap.add_argument("--kinds", default="random,extreme,close,walls,ellipsoids,meshwalls")
    args = ap.parse_args()
    lib = _ffi.hip()
    import verify_fuzz
    for name in ("cube", "arch", "bunny", "shadows", "cubes", "soccer", "rulers", "ladder_paradox"):
        try:
            s = Scene.from_file(name)
        except Exception as e:
            print(f"{name}: {e}")
            continue
        t = Tally()
        for k in range(args.states):
            f = k / max(args.states - 1, 1)
            speed = 0.99 * f
            ang, el = 2.0 * math.pi * 3.0 * f, 0.6 * math.sin(2.0 * math.pi * 5.0 * f)
            s.set_camera((speed * math.cos(el) * math.sin(ang), speed * math.sin(el), speed * math.cos(el) * math.cos(ang)), 30.0 * f)
            s.update_objects()
            for raw, root in objects_of(s):
                t.add(lib, raw, root, s.params["interval"])
        print(t.line(name), flush=True)
    for kind in args.kinds.split(","):
        t = Tally()
        for seed in range(args.seeds):
            try:
                s, _ = verify_fuzz.build(kind, seed)
            except Exception:
                continue
            for raw, root in objects_of(s):
                t.add(lib, raw, root, s.params["interval"])
        print(t.line("fuzz " + kind), flush=True)


if __name__ == "__main__":
    main()
