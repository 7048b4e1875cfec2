#!/usr/bin/env python3
"""How often does one octree walk test the SAME triangle again?  (A triangle that overlaps k leaves is listed in all k; VERDICT r03
item 3, step 1.)  Counted by the oracle (rpt_oracle_stats.distinct_tri_tests / repeats_of_previous_leaf), host only.
usage: python tools/repeated_tri_tests.py > profiles/r04_repeated_triangle_tests.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi                                               # noqa: E402
from conftest import load_config                                 # noqa: E402

print("one walk = one call of intersect_octree that passes the root box test (primary and shadow rays); the oracle's counters")
print("repeats = triangle tests of a triangle the same walk has tested before (skipping them is exact: opencl_kernel.cl:270 accepts")
print("0 <= dist < hit.dist with a non-increasing hit.dist, so a second application never changes anything)")
for name, (W, H), t in (("bunny", (1920, 1080), None), ("bunny", (3840, 2160), None), ("shadows", (1920, 1080), 16.0), ("shadows", (3840, 2160), 16.0)):
    s = load_config(name)
    if t is not None:
        s.set_camera((0, 0, 0), t)
        s.update_objects()
    _, _, st = oracle_ffi.render(s, W, H, want_rgb=False, want_stats=True, threads=os.cpu_count() or 8)
    tt, dt, rp, walks = st["tri_tests"], st["distinct_tri_tests"], st["repeats_of_previous_leaf"], st["root_aabb_hits"]
    print(f"{name:8s} {W}x{H}: walks {walks:9d}  leaf visits {st['leaf_visits']:9d}  triangle tests {tt:9d} ({tt / max(walks, 1):5.2f} per walk)  distinct {dt:9d}  "
          f"repeats {tt - dt:8d} = {100 * (tt - dt) / max(tt, 1):4.1f} %  |  repeats whose triangle was in the PREVIOUS leaf's list {rp:8d} = {100 * rp / max(tt, 1):4.1f} % of all tests")
