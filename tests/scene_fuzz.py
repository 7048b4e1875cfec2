"""Random scene descriptions for the fuzz tests (shared by GPU and CPU tests)."""
import numpy as np


def random_scene_text(rng):
    lines = []
    meshes = []
    if rng.random() < 0.7:
        meshes.append(rng.choice(["Models/pear.obj", "Models/bunny.obj", "Models/cube.obj", "Models/triangle.obj"]))
        if rng.random() < 0.3:
            meshes.append(rng.choice(["Models/cube.obj", "Models/pear.obj"]))
    for m in meshes:
        lines.append("M" + m)
    textures = []
    for t in ["Textures/box.jpg", "Textures/tile.jpg", "Textures/meterstick.png", "Textures/soccer.jpg"]:
        if rng.random() < 0.4:
            textures.append(t)
            lines.append("T" + t)
    n_obj = int(rng.integers(1, 9))
    has_textured_sphere = False
    for k in range(n_obj):
        kind = rng.choice(["s", "c", "m"] if meshes else ["s", "c"], p=[0.3, 0.4, 0.3] if meshes else [0.45, 0.55])
        if kind == "m":
            mi = int(rng.integers(0, len(meshes)))
            lines.append(f"Om{mi}")
            scale = float(rng.choice([1.0, 2.0, 12.0])) if "bunny" in meshes[mi] else float(rng.uniform(0.5, 2.0))
        else:
            lines.append("O" + kind)
            scale = float(rng.uniform(0.3, 2.5))
        pos = rng.uniform([-6, -4, -3], [6, 4, 14])
        if rng.random() < 0.1:
            pos = rng.uniform(-0.3, 0.3, size=3)          # camera inside / touching the object
        ang = float(rng.choice([0.0, rng.uniform(-3, 3)]))
        axis = rng.normal(size=3)
        sc = scale * rng.uniform(0.5, 1.5, size=3) if rng.random() < 0.5 else np.full(3, scale)
        lines.append(" p" + ",".join(f"{v:.4f}" for v in [*pos, ang, *axis, *sc]))
        lines.append(" c" + ",".join(f"{v:.3f}" for v in rng.uniform(0.05, 1.5, size=3)))
        if textures and rng.random() < 0.5:
            lines.append(f" t{int(rng.integers(0, len(textures)))}")
            has_textured_sphere = has_textured_sphere or kind == "s"
        if rng.random() < 0.3:
            lines.append(" l1")
        if rng.random() < 0.35:
            v = rng.normal(size=3)
            v = v / np.linalg.norm(v) * rng.choice([0.2, 0.6, 0.9, 0.97])
            lines.append(" v" + ",".join(f"{c:.5f}" for c in v))
        if rng.random() < 0.25:
            lines.append(f" f{rng.uniform(0.5, 3):.3f},{rng.uniform(0.1, 1.5):.3f}")
    lines.append(f"A{rng.uniform(0, 1):.3f}")
    if rng.random() < 0.5:
        lines.append("W" + ",".join(f"{v:.2f}" for v in rng.uniform(0.5, 6, size=3)))
    if rng.random() < 0.25:
        lines.append("I")
    lines.append("R")
    return "\n".join(lines) + "\n", has_textured_sphere


def extreme_scene_text(rng):
    """Objects at 0.5c .. 0.9999c in random directions, some of them far away or tiny (the camera then stands hundreds or
    thousands of radii away in the object's own frame), lights among them, light propagation on or off: the regime where the
    kernel's float arithmetic itself gets coarse (the sphere's discriminant, the boosted null direction) and every
    conservative cull has to allow for that."""
    lines = ["MModels/cube.obj"]
    for k in range(int(rng.integers(1, 4))):
        kind = rng.choice(["s", "c", "m0"])
        lines.append("O" + kind)
        far = rng.choice([1.0, 1.0, 10.0, 60.0])
        pos = rng.uniform([-8, -5, -6], [8, 5, 20]) * far
        sc = rng.uniform(0.3, 3.0, size=3) * rng.choice([0.02, 0.3, 1.0, 1.0, 5.0])
        ang = float(rng.uniform(-3, 3))
        axis = rng.normal(size=3)
        lines.append(" p" + ",".join(f"{v:.4f}" for v in [*pos, ang, *axis, *sc]))
        lines.append(" c1,1,1")
        if rng.random() < 0.3:
            lines.append(" l1")
        if rng.random() < 0.85:
            v = rng.normal(size=3)
            v = v / np.linalg.norm(v) * rng.choice([0.5, 0.9, 0.99, 0.999, 0.9999])
            lines.append(" v" + ",".join(f"{c:.6f}" for c in v))
    lines.append("A0.5")
    if rng.random() < 0.4:
        lines.append("I")
    lines.append("R")
    return "\n".join(lines) + "\n"


def close_scene_text(rng):
    """Large boxes, slabs, rulers and spheres CLOSE to the camera (a few of their own sizes away, some enclosing it) at 0 .. 0.99c,
    meshes among them, light propagation on or off: outlines that fill the screen, leave it, and pass near the camera — where the
    mapping from an outline's own parameter to the image is least uniform (the regime of the soak's seed 28 819)."""
    lines = ["MModels/cube.obj", "MModels/pear.obj"]
    for k in range(int(rng.integers(1, 5))):
        kind = rng.choice(["s", "c", "c", "m0", "m1"])
        lines.append("O" + kind)
        size = float(rng.choice([0.5, 2.0, 6.0, 15.0]))
        shape = rng.choice(["cube", "slab", "ruler"])
        sc = np.full(3, size) * rng.uniform(0.6, 1.4, size=3)
        if shape == "slab":
            sc[int(rng.integers(0, 3))] *= 0.02
        elif shape == "ruler":
            keep = int(rng.integers(0, 3))
            for a in range(3):
                if a != keep:
                    sc[a] *= 0.03
        direction = rng.normal(size=3)
        direction[2] = abs(direction[2]) * rng.choice([1.0, 1.0, 1.0, -1.0])
        pos = direction / np.linalg.norm(direction) * size * rng.uniform(0.3, 4.0)
        ang = float(rng.uniform(-3, 3))
        axis = rng.normal(size=3)
        lines.append(" p" + ",".join(f"{v:.4f}" for v in [*pos, ang, *axis, *sc]))
        lines.append(" c1,1,1")
        if rng.random() < 0.3:
            lines.append(" l1")
        if rng.random() < 0.7:
            v = rng.normal(size=3)
            v = v / np.linalg.norm(v) * rng.choice([0.3, 0.7, 0.9, 0.97, 0.99])
            lines.append(" v" + ",".join(f"{c:.6f}" for c in v))
    lines.append("A0.5")
    if rng.random() < 0.4:
        lines.append("I")
    lines.append("R")
    return "\n".join(lines) + "\n"


def walls_scene_text(rng):
    """Huge thin slabs, walls and beams a fraction of a unit from the camera, at any orientation, some of them moving, with and
    without light propagation: every corner at or behind the horizon while the object fills the screen (the case rpt_verify_frame's
    first sweep found in ladder_paradox.txt)."""
    lines = []
    n = int(rng.integers(1, 4))
    for _ in range(n):
        big = float(10.0 ** rng.uniform(1.0, 3.0))
        thin = float(10.0 ** rng.uniform(-3.0, -0.5))
        shape = rng.choice(["wall", "floor", "beam", "slab"])
        sc = {"wall": (big, big * float(rng.uniform(0.01, 1.0)), thin), "floor": (big, thin, big), "beam": (big, thin, thin * float(rng.uniform(1, 30))),
              "slab": (big * float(rng.uniform(0.05, 1)), big, thin)}[shape]
        d = float(10.0 ** rng.uniform(-1.5, 0.7))
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        pos = direction * d + rng.normal(size=3) * 0.3 * float(rng.random() < 0.5)
        axis = rng.normal(size=3)
        ang = float(rng.uniform(0, 6.28)) if rng.random() < 0.6 else 0.0
        lines.append("Oc")
        lines.append(" p%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f" % (pos[0], pos[1], pos[2], ang, axis[0], axis[1], axis[2], sc[0], sc[1], sc[2]))
        lines.append(" c%.3f,%.3f,%.3f" % tuple(rng.uniform(0.2, 1.0, size=3)))
        if rng.random() < 0.3:
            v = rng.normal(size=3)
            v = v / np.linalg.norm(v) * float(rng.choice([0.3, 0.9, 0.99]))
            lines.append(" v%.4f,%.4f,%.4f" % (v[0], v[1], v[2]))
    if rng.random() < 0.5:
        lines += ["Os", " l1", " p%.3f,%.3f,%.3f,0,0,1,0,0.2,0.2,0.2" % tuple(rng.normal(size=3) * 2), " c1,1,1"]
    lines.append("A%.3f" % rng.uniform(0.1, 0.9))
    if rng.random() < 0.5:
        lines.append("I")
    lines.append("R")
    return "\n".join(lines) + "\n"


def ellipsoids_scene_text(rng):
    """Spheres scaled into needles, discs and planets (scale ratios up to 1e5), a fraction of their smallest radius to hundreds of
    their largest away from the camera, rotated, some moving, some textured, with a light half of the time."""
    lines = []
    if rng.random() < 0.4:
        lines.append("TTextures/soccer.jpg")
    n = int(rng.integers(1, 4))
    for _ in range(n):
        base = float(10.0 ** rng.uniform(-2.0, 2.5))
        sc = [base * float(10.0 ** rng.uniform(-2.5, 0.0)) if rng.random() < 0.6 else base for _ in range(3)]
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        d = max(sc) * float(10.0 ** rng.uniform(-0.3, 1.5)) if rng.random() < 0.7 else min(sc) * float(rng.uniform(1.05, 4.0))
        pos = direction * d
        axis = rng.normal(size=3)
        ang = float(rng.uniform(0, 6.28)) if rng.random() < 0.7 else 0.0
        lines.append("Os")
        lines.append(" p%.5f,%.5f,%.5f,%.4f,%.4f,%.4f,%.4f,%.5f,%.5f,%.5f" % (pos[0], pos[1], pos[2], ang, axis[0], axis[1], axis[2], sc[0], sc[1], sc[2]))
        lines.append(" c%.3f,%.3f,%.3f" % tuple(rng.uniform(0.2, 1.0, size=3)))
        if lines[0].startswith("T") and rng.random() < 0.5:
            lines.append(" t0")
        if rng.random() < 0.4:
            v = rng.normal(size=3)
            v = v / np.linalg.norm(v) * float(rng.choice([0.3, 0.9, 0.99, 0.999]))
            lines.append(" v%.4f,%.4f,%.4f" % (v[0], v[1], v[2]))
    if rng.random() < 0.5:
        lines += ["Os", " l1", " p%.3f,%.3f,%.3f,0,0,1,0,0.2,0.2,0.2" % tuple(rng.normal(size=3) * 2), " c1,1,1"]
    lines.append("A%.3f" % rng.uniform(0.1, 0.9))
    if rng.random() < 0.4:
        lines.append("I")
    lines.append("R")
    return "\n".join(lines) + "\n"


def meshwalls_scene_text(rng):
    """One or two MESH objects (pear, cube.obj, triangle, bunny) scaled into slabs and needles next to the camera or around it, a
    light, sometimes a floor: the root box through the box bounds, the shadow-segment cull for meshes, walks that start inside."""
    meshes = [str(rng.choice(["Models/pear.obj", "Models/cube.obj", "Models/triangle.obj", "Models/bunny.obj"]))]
    lines = ["M" + m for m in meshes]
    n = int(rng.integers(1, 3))
    for _ in range(n):
        base = float(10.0 ** rng.uniform(-1.0, 2.5))
        sc = [base * float(10.0 ** rng.uniform(-3.0, 0.0)) if rng.random() < 0.6 else base for _ in range(3)]
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        d = float(10.0 ** rng.uniform(-1.5, 1.0))
        pos = direction * d
        axis = rng.normal(size=3)
        ang = float(rng.uniform(0, 6.28)) if rng.random() < 0.7 else 0.0
        lines.append("Om0")
        lines.append(" p%.5f,%.5f,%.5f,%.4f,%.4f,%.4f,%.4f,%.5f,%.5f,%.5f" % (pos[0], pos[1], pos[2], ang, axis[0], axis[1], axis[2], sc[0], sc[1], sc[2]))
        lines.append(" c%.3f,%.3f,%.3f" % tuple(rng.uniform(0.2, 1.0, size=3)))
        if rng.random() < 0.3:
            v = rng.normal(size=3)
            v = v / np.linalg.norm(v) * float(rng.choice([0.3, 0.9, 0.99]))
            lines.append(" v%.4f,%.4f,%.4f" % (v[0], v[1], v[2]))
    if rng.random() < 0.7:
        lines += ["Os", " l1", " p%.3f,%.3f,%.3f,0,0,1,0,0.2,0.2,0.2" % tuple(rng.normal(size=3) * 2), " c1,1,1"]
    if rng.random() < 0.4:
        lines += ["Oc", " p0,-2,0,0,0,1,0,50,0.1,50", " c0.5,0.5,0.5"]
    lines.append("A%.3f" % rng.uniform(0.1, 0.9))
    if rng.random() < 0.3:
        lines.append("I")
    lines.append("R")
    return "\n".join(lines) + "\n"
