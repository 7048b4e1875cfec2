#!/usr/bin/env python3
"""Soundness soak of csrc/rpt_bounds_certify.hpp against the oracle (host only, no GPU): for every object of the scenes of
tests/scene_fuzz.py's generators, a few dozen claims — the proposer's, copies of it pulled in and shifted, random rectangles and
octagons — are handed to rpt_certify_screen_bounds; every claim it PROVES must contain every pixel the oracle hits (the test
functions of tests/test_bounds_certificate.py, over a seed range).
usage: python tools/certify_soak.py --first 0 --last 2000 [--kinds random,extreme,...] [--jobs 6]"""
import argparse
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def work(job):
    kind, first, last = job
    import numpy as np
    import test_bounds_certificate as tbc
    import verify_fuzz
    acc = rej = scenes = 0
    bad = []
    for seed in range(first, last):
        try:
            scene, text = verify_fuzz.build(kind, seed)
        except Exception:
            continue
        scenes += 1
        try:
            a, r = tbc._soundness(scene, f"{kind} {seed}", np.random.default_rng(seed), frames=((160, 90), (96, 128)))
            acc, rej = acc + a, rej + r
        except AssertionError as e:
            bad.append(str(e)[:400])
    return kind, scenes, acc, rej, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--last", type=int, default=1000)
    ap.add_argument("--kinds", default="random,extreme,close,walls,ellipsoids,meshwalls")
    ap.add_argument("--jobs", type=int, default=6)
    args = ap.parse_args()
    jobs = []
    step = max(1, (args.last - args.first) // (4 * args.jobs))
    for kind in args.kinds.split(","):
        for a in range(args.first, args.last, step):
            jobs.append((kind, a, min(a + step, args.last)))
    tot = {}
    failures = 0
    with mp.Pool(args.jobs) as pool:
        for kind, scenes, acc, rej, bad in pool.imap_unordered(work, jobs):
            t = tot.setdefault(kind, [0, 0, 0, 0])
            t[0] += scenes; t[1] += acc; t[2] += rej; t[3] += len(bad)
            failures += len(bad)
            for b in bad:
                print("UNSOUND:", b, flush=True)
    for kind, t in tot.items():
        print(f"{kind:11s} seeds {args.first}..{args.last - 1}: {t[0]} scenes, {t[1]} claims proven (all contain every oracle hit), {t[2]} rejected, {t[3]} unsound", flush=True)
    print(f"TOTAL unsound: {failures}")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
