"""Parity of the HIP render path (through the C-ABI) against the CPU oracle.

Bar (BASELINE.json north_star): <= 1e-4 max per-channel deviation on float RGB.  The HIP kernel
restates the oracle's fp32 operation order without contraction, so what is actually asserted is
stronger: float RGB bit-identical and packed bytes identical on EVERY scene, textured spheres included
(asin/atan2 are the same explicit fdlibm algorithm in plain IEEE operations on both sides).  The opt-in
relaxed-arithmetic variants 50/51 are not parity paths and are not in this file (tests/test_gpu_relaxed.py).
"""
import numpy as np
import pytest

import oracle_ffi
from conftest import CONFIGS, load_config

pytestmark = pytest.mark.gpu

TOL = 1e-4   # the north_star tolerance, per channel, on tonemapped float RGB


@pytest.fixture(scope="module")
def renderer():
    from relativitypathtracer_amd.renderer import Renderer
    r = Renderer(0)
    yield r
    r.close()


def _render_gpu(r, scene, W, H, variant=0):
    r.set_variant(variant)
    r.upload_scene(scene)
    r.set_scene_params(scene, W, H)
    r.set_rows(0, 1, False)
    r.set_output(None)
    r.set_debug_rgb(True)
    r.render()
    return r.read_framebuffer(), r.read_debug_rgb()


SIZES = {"cube": (640, 480), "arch": (480, 270), "arch_t0": (480, 270), "bunny": (480, 270), "shadows": (480, 270),
         "cubes": (480, 270), "rulers": (480, 270), "ladder": (480, 270), "soccer": (320, 184)}
EXACT = {"cube", "arch", "arch_t0", "bunny", "shadows", "cubes", "rulers", "ladder", "soccer"}   # every scene: textured spheres too (same explicit asin/atan2 on both sides)


VARIANTS = [0, 1, 3, 41, 43, 44]   # include/rpt.h: 0 = default (41 async / 43 blocking / 44 mesh-free); 1 = reference-layout kernel; 3 = derived layouts, no culling (what rpt_verify_frame compares with).  The measurement arms live in librpt_hip_diag.so: tests/test_gpu_diag_arms.py


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", list(SIZES))
def test_frame_matches_oracle(renderer, name, variant):
    W, H = SIZES[name]
    scene = load_config(name)
    px, rgb = _render_gpu(renderer, scene, W, H, variant)
    opx, orgb, _ = oracle_ffi.render(scene, W, H)
    assert np.array_equal(px["x"], opx["x"]) and np.array_equal(px["y"], opx["y"])
    assert np.all(px["rgba"][:, 3] == 1)
    err = float(np.max(np.abs(rgb - orgb)))
    assert err <= TOL, f"{name}: max |rgb - oracle| = {err}"
    if name in EXACT:
        assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32)), f"{name}: float RGB not bit-identical"
        assert np.array_equal(px["rgba"], opx["rgba"]), f"{name}: packed bytes differ"
    else:
        d = np.abs(px["rgba"].astype(np.int16) - opx["rgba"].astype(np.int16))
        assert d.max() <= 1, f"{name}: packed bytes differ by more than one LSB"


@pytest.mark.parametrize("msaa", [2, 3])
@pytest.mark.parametrize("name", ["bunny", "shadows", "arch", "cubes", "soccer"])
def test_msaa_matches_oracle(name, msaa):
    """MSAASAMPLES (opencl_kernel.cl:7, :641-648) other than the shipped 1: rpt_set_msaa's kernels — n x n rays per pixel at
    (x + i/n, y + j/n), summed in the reference's order, a miss contributing the background, divided by n^2 before the tonemap —
    against the oracle's restatement of the same loop, float RGB bit for bit, culled (46) and un-culled (47), on a frame whose
    size is no multiple of the tile; the cull changes nothing (rpt_verify_frame).  (No reference output exists for n > 1: what
    pins these frames to the reference is the one-sample path.)"""
    from relativitypathtracer_amd.renderer import Renderer, RenderError
    W, H = (333, 190) if name != "soccer" else (250, 141)
    scene = load_config(name)
    opx, orgb, _ = oracle_ffi.render(scene, W, H, msaa=msaa)
    opx1, _, _ = oracle_ffi.render(scene, W, H, want_rgb=False)
    assert not np.array_equal(opx["rgba"], opx1["rgba"])            # more samples do change edges
    r = Renderer(0)
    try:
        r.set_msaa(msaa)
        for variant, kernel in ((0, 46), (3, 47)):
            px, rgb = _render_gpu(r, scene, W, H, variant)
            assert r.last_variant() == kernel
            assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32)), f"{name} msaa {msaa} variant {variant}: float RGB not bit-identical"
            assert np.array_equal(px["rgba"], opx["rgba"]) and np.array_equal(px["x"], opx["x"]) and np.array_equal(px["y"], opx["y"])
        r.set_variant(0)
        assert r.verify_frame() == 0
        r.set_variant(1)
        with pytest.raises(RenderError):
            r.render()                       # the reference-layout kernel has no multi-sample form
        r.set_variant(0)
        with pytest.raises(RenderError):
            r.set_msaa(0)
        r.set_msaa(1)
        px, _ = _render_gpu(r, scene, W, H, 0)
        assert np.array_equal(px["rgba"], opx1["rgba"]) and r.last_variant() in (43, 44)
    finally:
        r.close()


def test_camera_inside_the_mesh_root_box(renderer):
    """What Screenshots/mesh3.png shows and nothing the reference ships can pin (no analytic object in view, a model the tree
    lacks): the camera next to and INSIDE the bunny's root box, so that primary rays start in the walk's origin-inside-the-root
    branch (opencl_kernel.cl:233-248: descend to the leaf that holds the origin, then walk).  HIP == oracle bit for bit through
    every kernel, from a moving camera (that is the only way the reference's camera gets anywhere: position = gamma v T), with
    the oracle's own counter saying that the branch was taken by most primary rays; the screen bounds give such an object the
    full plane, and the culled kernels agree with the un-culled one."""
    import math
    from relativitypathtracer_amd import Scene
    scene = Scene.from_file("bunny")
    objs = scene.objects()
    mesh = int(np.flatnonzero(np.asarray(objs["type"]) == 2)[0])
    node = scene.octrees()[int(objs["meshIndex"][mesh])]
    M = np.array(objs[mesh]["M"], dtype=np.float64).reshape(4, 4)
    centre = M[:3, :3] @ (0.5 * (np.array(node["min"][:3], dtype=np.float64) + np.array(node["max"][:3], dtype=np.float64))) + M[:3, 3]
    W, H = 480, 270
    inside_seen = 0
    for frac, speed in ((1.0, 0.5), (0.8, 0.3), (1.15, 0.7)):
        target = centre * frac
        d = target / np.linalg.norm(target)
        gamma = 1.0 / math.sqrt(1.0 - speed * speed)
        scene.set_camera(tuple(float(x) for x in d * speed), float(np.linalg.norm(target) / (gamma * speed)))
        scene.update_objects()
        opx, orgb, st = oracle_ffi.render(scene, W, H, want_stats=True)
        inside_seen += int(st["inside_starts"] > 0.5 * W * H)
        for variant in (0, 1, 3, 41, 43):
            px, rgb = _render_gpu(renderer, scene, W, H, variant)
            assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32)), f"frac {frac} variant {variant}: float RGB not bit-identical"
            assert np.array_equal(px["rgba"], opx["rgba"]), f"frac {frac} variant {variant}"
        renderer.set_variant(0)
        assert renderer.verify_frame() == 0
    assert inside_seen >= 1


def test_frames_wider_than_the_proven_window_render_unculled(renderer):
    """The screen bounds are proven for the window |u| <= 2, |v| <= 1/2 (csrc/rpt_bounds_certify.hpp): every pixel of a frame of at
    most 4 : 1.  A wider frame has pixels outside it, and launch() gives it the un-culled kernel — same pixels as the oracle."""
    scene = load_config("shadows")
    for (W, H, kernels) in ((1600, 400, (43,)), (1608, 400, (3,)), (2000, 250, (3,)), (800, 200, (43,))):
        px, rgb = _render_gpu(renderer, scene, W, H, 0)
        assert renderer.last_variant() in kernels, (W, H, renderer.last_variant())
        opx, orgb, _ = oracle_ffi.render(scene, W, H)
        assert np.array_equal(px["rgba"], opx["rgba"]) and np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32)), (W, H)
        for variant in (41, 43):          # asked for explicitly on a frame they are not proven for: the un-culled kernel all the same
            _render_gpu(renderer, scene, W, H, variant)
            assert renderer.last_variant() == (variant if kernels != (3,) else 3), (W, H, variant, renderer.last_variant())
    renderer.set_variant(0)


def test_odd_resolution_guard(renderer):
    """Width/height that are not multiples of the 32x8 strip: the reference has no bounds guard."""
    scene = load_config("shadows")
    W, H = 333, 77
    px, rgb = _render_gpu(renderer, scene, W, H)
    opx, orgb, _ = oracle_ffi.render(scene, W, H)
    assert np.array_equal(px["rgba"], opx["rgba"])
    assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32))


def test_tile_culling_never_drops_a_hit(renderer):
    """The default kernels skip objects per 8x8 tile by each wavefront's own lane-parallel test of per-object image-plane
    rectangles + __ballot (41, 43); the plain kernel (variant 3) tests every object for every pixel; rpt_verify_frame makes the
    same comparison on the device.  Frames must be identical for arbitrary camera
    velocities, camera times, object velocities and resolutions (incl. coarse ones where tiles are wide)."""
    from relativitypathtracer_amd import Scene
    rng = np.random.default_rng(2024)
    scenes = ["shadows", "bunny", "arch", "cubes", "rulers", "ladder_paradox", "soccer", "cube"]
    sizes = [(480, 270), (333, 77), (160, 120), (64, 48), (1280, 720)]
    checked = 0
    for trial in range(160):
        name = scenes[trial % len(scenes)]
        W, H = sizes[trial % len(sizes)]
        s = Scene.from_file(name)
        v = rng.normal(size=3)
        v = v / np.linalg.norm(v) * rng.choice([0.0, 0.3, 0.9, 0.99])
        s.set_camera(tuple(float(c) for c in v), float(rng.uniform(-5, 25)))
        s.update_objects()
        frames = []
        for variant in (3, 41, 43):
            px, rgb = _render_gpu(renderer, s, W, H, variant)
            frames.append((px, rgb))
        renderer.set_variant(0)
        assert renderer.verify_frame() == 0, f"{name} {W}x{H} v={v}: rpt_verify_frame reports culled != un-culled"
        for other in frames[1:]:
            assert np.array_equal(frames[0][0]["rgba"], other[0]["rgba"]), f"{name} {W}x{H} v={v}: packed bytes differ"
            assert np.array_equal(frames[0][1].view(np.uint32), other[1].view(np.uint32))
        checked += int((frames[0][0]["rgba"][:, :3] != frames[0][0]["rgba"][0, :3]).any())
    assert checked > 80     # most trials actually had something on screen
