#!/usr/bin/env python3
"""Synthetic denser stand-in for the reference's missing StanfordBunny.obj: Models/bunny.obj with every
triangle split 1 -> 4 `levels` times (midpoints, no smoothing).  NON-REFERENCE data: used only as an extra, clearly
labelled data point for larger octrees.  Writes an OBJ with 6-decimal coordinates."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def subdivide_obj(src, dst, levels=2):
    V, F = [], []
    for line in open(src):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            V.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            F.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
    V, F = np.array(V, dtype=np.float64), np.array(F)
    for _ in range(levels):
        edge, new_v, nf = {}, list(V), []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in edge:
                edge[k] = len(new_v)
                new_v.append((V[a] + V[b]) / 2)
            return edge[k]
        for a, b, c in F:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
        V, F = np.array(new_v), np.array(nf)
    with open(dst, "w") as f:
        f.write("# synthetic: Models/bunny.obj subdivided 1->4 x%d (non-reference data)\n" % levels)
        for v in V:
            f.write("v %.6f %.6f %.6f\n" % tuple(v))
        for a, b, c in F:
            f.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
    return len(V), len(F)


def dense_bunny_scene(workdir, levels=2):
    """A Scene equal to Scenes/bunny.txt with the mesh replaced by the subdivided one (texture kept)."""
    sys.path.insert(0, ROOT)
    from relativitypathtracer_amd import Scene
    from relativitypathtracer_amd.scene import ASSET_ROOT
    os.makedirs(os.path.join(workdir, "Models"), exist_ok=True)
    dst = os.path.join(workdir, "Models", f"bunny_x{4 ** levels}.obj")
    if not os.path.exists(dst):
        subdivide_obj(os.path.join(ASSET_ROOT, "Models", "bunny.obj"), dst, levels)
    text = open(os.path.join(ASSET_ROOT, "Scenes", "bunny.txt")).read()
    text = text.replace("MModels/StanfordBunny.obj", "M" + dst).replace("TTextures/", "T" + os.path.join(ASSET_ROOT, "Textures") + "/")
    s = Scene(asset_root="/")
    s.inputScene(text)
    return s


if __name__ == "__main__":
    import time
    sys.path.insert(0, ROOT)
    from relativitypathtracer_amd.renderer import Renderer
    t = time.time()
    s = dense_bunny_scene("/tmp/rpt_dense", int(sys.argv[1]) if len(sys.argv) > 1 else 2)
    d = s.desc()
    print(f"scene built in {time.time()-t:.2f} s: {d.triangle_words//9} triangles, {d.octree_count} nodes, {d.octree_tri_count} octreeTris")
    s.set_camera((0, 0, 0), 0.0)
    s.update_objects()
    r = Renderer(0)
    for W, H in [(1920, 1080), (3840, 2160)]:
        r.upload_scene(s); r.set_scene_params(s, W, H); r.set_output(None)
        ms = r.timed_frames(20)
        print(f"dense bunny {W}x{H}: {ms:.4f} ms  {W*H/ms/1e3:.1f} Mrays/s")
