#!/usr/bin/env python3
"""The WRONG screen bounds of rounds 1-3, as fixtures.  Every soak find of the first three rounds was an outline that the sampling of
csrc/rpt_screen_bounds.hpp had not followed: the bounds it proposed left out pixels the kernel hits.  This script rebuilds those
proposals from the repository's own history — the header as it stood in the commit BEFORE each fix (git show), compiled into a
throw-away program — for the scenes the finds are kept as in tests/test_screen_bounds.py, keeps every (object, bounds) pair for
which the oracle shows hit pixels OUTSIDE the bounds, and writes them to tests/golden/historic_wrong_bounds.json.
tests/test_bounds_certificate.py hands each of them to rpt_certify_screen_bounds: the proof must FAIL for every one.
Needs the git history and the oracle: run in the development container, `python tests/golden/make_historic_claims.py`."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_ffi                                               # noqa: E402,F401
import test_screen_bounds as tsb                                 # noqa: E402
import verify_fuzz                                               # noqa: E402
from relativitypathtracer_amd import Scene                       # noqa: E402
from scene_fuzz import extreme_scene_text                        # noqa: E402

PROGRAM = r"""
#include <cstdio>
#include "rpt_screen_bounds.hpp"
int main(int argc, char **argv) {
    FILE *f = std::fopen(argv[1], "rb");
    int interval = 0, has_root = 0;
    rpt_object o;
    float root[6];
    if (!f || std::fread(&interval, 4, 1, f) != 1 || std::fread(&has_root, 4, 1, f) != 1 || std::fread(root, 4, 6, f) != 6 || std::fread(&o, sizeof o, 1, f) != 1) return 2;
    const rptb::Rect r = rptb::object_rect(o, interval, has_root ? root : nullptr);
    std::printf("%.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", r.u0, r.v0, r.u1, r.v1, r.p_lo, r.p_hi, r.m_lo, r.m_hi);
    return 0;
}
"""


def old_program(commit, tmp):
    """object_rect of csrc/rpt_screen_bounds.hpp as of `commit`, as an executable."""
    d = os.path.join(tmp, commit.replace("^", "_parent"))
    os.makedirs(os.path.join(d, "relativitypathtracer_amd", "csrc"), exist_ok=True)
    os.makedirs(os.path.join(d, "include"), exist_ok=True)
    for path in ("relativitypathtracer_amd/csrc/rpt_screen_bounds.hpp", "include/rpt_layout.h"):
        text = subprocess.run(["git", "show", f"{commit}:{path}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout
        with open(os.path.join(d, path), "w") as f:
            f.write(text)
    src = os.path.join(d, "relativitypathtracer_amd", "csrc", "main.cpp")
    with open(src, "w") as f:
        f.write(PROGRAM)
    exe = os.path.join(d, "old_rect")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, src], check=True)
    return exe


def extreme(seed):
    rng = np.random.default_rng(550000 + seed)
    scene = Scene()
    scene.inputScene(extreme_scene_text(rng))
    v = rng.normal(size=3)
    v = v / np.linalg.norm(v) * rng.choice([0.0, 0.5, 0.9, 0.99, 0.999])
    scene.set_camera(tuple(float(c) for c in v), float(rng.uniform(-5, 40)))
    scene.update_objects()
    return scene


def random_scene(seed):
    from scene_fuzz import random_scene_text
    rng = np.random.default_rng(7000 + seed)
    text, _ = random_scene_text(rng)
    scene = Scene()
    scene.inputScene(text)
    vel = rng.normal(size=3)
    vel = vel / np.linalg.norm(vel) * rng.choice([0.0, 0.0, 0.5, 0.95])
    scene.set_camera(tuple(float(c) for c in vel), float(rng.uniform(-3, 20)))
    scene.update_objects()
    return scene


# (label, the commit whose PARENT still had the flaw, scene, frames to look for lost pixels at)
FINDS = [
    ("uniform segments: extreme 28819", "73ec40c", lambda: extreme(28819), [(320, 184), (160, 90)]),
    ("margins from an estimate: extreme 7100", "73ec40c", lambda: extreme(7100), [(320, 184), (160, 90)]),
    ("corners and midpoint behind the camera: random 1919", "73ec40c", lambda: random_scene(1919), [(160, 90), (128, 96), (200, 80)]),
    ("stretches off the screen skipped: close 755", "2cb2281", lambda: tsb.close_scene(755), [(160, 90), (200, 80), (320, 180)]),
    ("visible end inside one sixteenth: close 8660", "2cb2281", lambda: tsb.close_scene(8660), [(160, 90), (200, 80), (320, 180)]),
    ("visible end inside one sixteenth: close 20897", "2cb2281", lambda: tsb.close_scene(20897), [(160, 90), (200, 80), (320, 180)]),
    ("wall with every corner at the horizon: ladder_paradox sweep 311", "e81f89c", lambda: tsb._sweep_state("ladder_paradox", 311), [(320, 180)]),
    ("beam with both ends behind the camera: walls 1598", "57e0bce", lambda: verify_fuzz.build("walls", 1598)[0], [(333, 77)]),
    ("beam with both ends behind the camera: walls 7396", "57e0bce", lambda: verify_fuzz.build("walls", 7396)[0], [(180, 320)]),
    ("beam with both ends behind the camera: walls 9588", "57e0bce", lambda: verify_fuzz.build("walls", 9588)[0], [(180, 320)]),
    ("beam with both ends behind the camera: walls 17216", "57e0bce", lambda: verify_fuzz.build("walls", 17216)[0], [(320, 184)]),
    ("beam with both ends behind the camera: walls 26583", "57e0bce", lambda: verify_fuzz.build("walls", 26583)[0], [(320, 180)]),
] + [(f"float sphere larger than the exact one: extreme {s}", "2757a62", (lambda s=s: extreme(s)), [(160, 90)]) for s in range(0, 400)]


def main():
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        programs = {}
        for label, commit, build, frames in FINDS:
            parent = commit + "^"
            if parent not in programs:
                programs[parent] = old_program(parent, tmp)
            scene = build()
            objs, nodes = scene.objects(), scene.octrees()
            interval = scene.params["interval"]
            for i in range(min(len(objs), 64)):
                root = np.zeros(6, dtype=np.float32)
                has_root = int(objs["type"][i]) == 2
                if has_root:
                    n = nodes[int(objs["meshIndex"][i])]
                    root[:3], root[3:] = n["min"][:3], n["max"][:3]
                rec = os.path.join(tmp, "object.bin")
                with open(rec, "wb") as f:
                    f.write(np.array([interval, int(has_root)], dtype=np.int32).tobytes() + root.tobytes() + objs[i:i + 1].tobytes())
                b = tuple(float(x) for x in subprocess.run([programs[parent], rec], capture_output=True, text=True, check=True).stdout.split())
                lost_best = None
                for (W, H) in frames:
                    ys, xs = np.mgrid[0:H, 0:W]
                    u, v = (xs / W - 0.5) * (W / H), ys / H - 0.5
                    lost = int((tsb.hit_mask(scene, i, W, H) & ~tsb.inside_bounds(b, u, v)).sum())
                    if lost and (lost_best is None or lost > lost_best[0]):
                        lost_best = (lost, W, H)
                if lost_best:
                    out.append({"find": label, "header_commit": subprocess.run(["git", "rev-parse", "--short", parent], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
                                "object_index": i, "interval": interval, "root_bounds": [float(x) for x in root] if has_root else None,
                                "object_hex": objs[i:i + 1].tobytes().hex(), "bounds": list(b), "lost_pixels": lost_best[0], "frame": list(lost_best[1:])})
                    print(f"{label}: object {i}: {lost_best[0]} hit pixels outside the bounds of {parent} at {lost_best[1]}x{lost_best[2]}", flush=True)
    path = os.path.join(HERE, "historic_wrong_bounds.json")
    with open(path, "w") as f:
        json.dump({"note": "screen bounds proposed by earlier versions of csrc/rpt_screen_bounds.hpp that lose hit pixels (made by make_historic_claims.py from the git history)",
                   "claims": out}, f, indent=1)
    print(f"{len(out)} wrong claims -> {path}")


if __name__ == "__main__":
    main()
