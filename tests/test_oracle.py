"""Pins of the CPU oracle (oracle/rpt_oracle.c) — what stands between it and the reference.

The reference has no tests, golden vectors or fixtures for this path and cannot be built in this
image, so the oracle is pinned by what the reference DOES ship or recorded:

0. (the mesh / octree path) its four grabs of Scenes/shadows.txt — shadows1/2/4/5.png, README.md:117-122 — a pear MESH lit
   by a light crossing the scene at 0.95c: OBJ loader, octree builder, intersect_octree / intersect_triangle /
   intersect_AABB / getOppositeBoxSide for primary and shadow rays.  Only the clock is unknown; at the recovered
   milliseconds <= 55 of 3.5 M pixels are off by more than 1 LSB and at most ONE of the pear's own 34 099 pixels;
1. its own output images (Screenshots/*.png — cut into the fixtures tests/golden/ref_*.png by
   tests/golden/make_reference_fixtures.py): the static scenes cube1.png and arch1.png (<= 1 LSB on every pixel of
   arch1), and the MOVING-camera grabs cube2.png, cube3.png (0.9c, without / with light propagation) and arch2.png
   (0.95c towards the arch), whose unrecorded velocity and clock were recovered by
   tests/golden/fit_reference_camera.py — arch2 is reproduced to <= 1 LSB on all but 4 of 3.5 M pixels;
2. the per-ray work counts SURVEY.md §8(a) recorded from the reference run during the survey;
3. committed golden frames of the oracle itself (drift detector).
"""
import math
import os

import numpy as np
import pytest
from PIL import Image

import oracle_ffi
from conftest import (CLIENT_H, CLIENT_W, CONFIGS, check_mesh1, REFERENCE_GIF_FRAMES, REFERENCE_SHOTS, SHADOWS_CROP, SHADOWS_PEAR_OBJECT,
                      load_config, load_reference_shot)
from relativitypathtracer_amd import Scene

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _render_top_down(shot, rows=(0, CLIENT_H), scene=None, dt=0.0):
    """Oracle render of a reference screenshot's camera state; client rows [rows) top-down as int16 RGB."""
    c = REFERENCE_SHOTS[shot]
    s = scene or load_reference_shot(shot)
    if dt:
        s.set_camera(c["v"], c["t"] + dt)
        s.update_objects()
    y0, y1 = rows
    px, _, _ = oracle_ffi.render(s, CLIENT_W, CLIENT_H, rows=(CLIENT_H - y1, CLIENT_H - y0), want_rgb=False)
    return px["rgba"].reshape(CLIENT_H, CLIENT_W, 4)[CLIENT_H - y1:CLIENT_H - y0][::-1, :, :3].astype(np.int16)   # framebuffer row 0 is the bottom row


def _load(name):
    return np.asarray(Image.open(os.path.join(GOLDEN, name)).convert("RGB")).astype(np.int16)


def test_reference_screenshot_arch1():
    """Scenes/arch.txt, stationary: light, 4 cubes, shadows, 1024^2 texture, Hable tonemap, packing."""
    img = _render_top_down("arch1")
    d = np.abs(img[::4, ::4] - _load("ref_arch1_stride4.png"))
    assert d.max() <= 1, f"stride-4 subsample: max byte difference {d.max()}"
    assert (d > 0).mean() < 0.01
    crop = np.abs(img[300:700, 960:1600] - _load("ref_arch1_crop_y300_x960.png"))
    assert crop.max() <= 1, f"full-resolution crop: max byte difference {crop.max()}"


def test_reference_screenshot_cube1():
    """Scenes/cube.txt, stationary, light propagation off: textured cube (uv convention, bilinear fetch).

    JPEG decoders differ by an LSB and the cube's silhouette/texture seams are one-pixel features, so a
    few pixels differ; everything else must agree to 2 levels."""
    img = _render_top_down("cube1")
    d = np.abs(img[::4, ::4] - _load("ref_cube1_stride4.png")).max(axis=2)
    assert (d > 2).mean() < 5e-4, f"fraction of pixels off by more than 2 levels: {(d > 2).mean()}"
    assert np.abs(img[::4, ::4] - _load("ref_cube1_stride4.png")).mean() < 0.02


def test_reference_screenshot_arch2_moving_camera():
    """Scenes/arch.txt from a camera moving towards the arch at 0.95c (README.md:81-83): per-object Lorentz
    boosts, aberration, light-delayed positions, light-frame shading and shadow rays, all at once.  At the
    recovered state (REFERENCE_SHOTS) the reference's grab is reproduced to <= 1 LSB on all but 4 of its
    3.5 M pixels; one millisecond of camera clock away, a tenth of the brick floor is wrong."""
    img = _render_top_down("arch2")
    d = np.abs(img[::4, ::4] - _load("ref_arch2_stride4.png")).max(axis=2)
    assert (d > 1).sum() <= 4, f"stride-4 subsample: {(d > 1).sum()} pixels off by more than 1 level"
    assert (d > 0).mean() < 0.02
    ref_crop = _load("ref_arch2_crop_y900_x960.png")
    crop = np.abs(img[900:1300, 960:1600] - ref_crop)
    assert crop.max() <= 1, f"full-resolution crop of the brick floor: max byte difference {crop.max()}"
    off = _render_top_down("arch2", rows=(900, 1300), dt=0.001)[:, 960:1600]
    assert (np.abs(off - ref_crop).max(axis=2) > 1).mean() > 0.05       # the pin is sharp: 1 ms off is visible


@pytest.mark.parametrize("shot,crop,fixture", [("cube2", (826, 1377, 1150, 1410), "ref_cube2_crop_y826_x1150.png"),
                                               ("cube3", (826, 1377, 1000, 1620), "ref_cube3_crop_y826_x1000.png")])
def test_reference_screenshot_cube_moving_camera(shot, crop, fixture):
    """Scenes/cube.txt from a camera moving right at 0.9c (README.md:91-94): cube2 without light propagation
    (length contraction on the simultaneity slice), cube3 with it (the retarded, Terrell-rotated view).
    Both grabs come out at ONE velocity, tanh(7373/5000) c.  The < 0.1 % of crate pixels that differ are
    texels (box.jpg: CImg/libjpeg there, Pillow here) and silhouette pixels, as for the static cube1."""
    y0, y1, x0, x1 = crop
    img = _render_top_down(shot)
    d = np.abs(img[::4, ::4] - _load(f"ref_{shot}_stride4.png")).max(axis=2)
    assert (d > 2).mean() < 5e-4, f"stride-4 subsample: fraction off by more than 2 levels {(d > 2).mean()}"
    ref_crop = _load(fixture)
    dc = np.abs(img[y0:y1, x0:x1] - ref_crop).max(axis=2)
    assert (dc > 1).sum() <= 300, f"full-resolution crop: {(dc > 1).sum()} pixels off by more than 1 level"
    if shot == "cube3":     # the pin is sharp: one millisecond of clock later a quarter of the crate is wrong
        off = _render_top_down(shot, rows=(y0, y1), dt=0.001)[:, x0:x1]
        assert (np.abs(off - ref_crop).max(axis=2) > 1).sum() > 20000


def test_reference_screenshot_sphere_stationary_textured_sphere():
    """Screenshots/sphere_stationary.png (README.md:124-125): a textured sphere at rest — sphere (u,v) through atan2 and
    asin (opencl_kernel.cl:356-357) and the bilinear fetch.  The grab was taken with the ball turned by 2 rad about y
    (REFERENCE_SHOTS); with that, every pixel of the grab is within 1 LSB, and 0.004 rad away 40 000 are not."""
    img = _render_top_down("sphere_stationary")
    d = np.abs(img[::4, ::4] - _load("ref_sphere_stationary_stride4.png"))
    assert d.max() <= 1, f"stride-4 subsample: max byte difference {d.max()}"
    ref_crop = _load("ref_sphere_stationary_crop_y380_x960.png")
    crop = np.abs(img[380:1020, 960:1600] - ref_crop)
    assert crop.max() <= 1, f"full-resolution crop of the ball: max byte difference {crop.max()}"
    # (a third of the ball's channel values differ by exactly one level: soccer.jpg decoded by CImg/libjpeg there, Pillow here)
    s = Scene()
    s.inputScene(REFERENCE_SHOTS["sphere_stationary"]["text"].replace("p0,0,5,2,", "p0,0,5,2.004,"))
    s.update_objects()
    off = _render_top_down("sphere_stationary", rows=(380, 1020), scene=s)[:, 960:1600]
    assert (np.abs(off - ref_crop).max(axis=2) > 1).sum() > 20000


def test_reference_screenshot_mesh1_the_headline_scene():
    """Screenshots/mesh1.png (README.md:85-87): Scenes/bunny.txt — the scene BASELINE.json's metric is quoted on — from a camera
    at rest.  Pinned pixel for pixel: the light sphere (every pixel of the grab's crop identical), the background and the framing.
    NOT pinned: the bunny's pixels — the grab shows Models/StanfordBunny.obj, which the reference tree lacks; Models/bunny.obj is
    the same model in the same pose under another normalisation (silhouette IoU 0.94 after a similarity of scale 1.28).  The mesh
    code path itself is pinned on Models/pear.obj by the four shadows grabs below."""
    img = _render_top_down("mesh1")
    iou, scale = check_mesh1(img, _load("ref_mesh1_stride4.png"), _load("ref_mesh1_crop_y300_x1230.png"))
    assert 1.2 < scale < 1.36
    # the pin is sharp: the light 0.002 units higher moves its outline
    s = Scene()
    s.inputScene(open(os.path.join(os.path.dirname(GOLDEN), "..", "assets", "reference", "Scenes", "bunny.txt")).read().replace("p0,2,4,", "p0,2.002,4,"))
    s.update_objects()
    off = _render_top_down("mesh1", rows=(300, 390), scene=s)[:, 1230:1330]
    assert np.abs(off - _load("ref_mesh1_crop_y300_x1230.png")).max() > 50


def test_reference_screenshot_mesh2_light_sphere_from_a_receding_camera():
    """Screenshots/mesh2.png: the bunny scene from a MOVING camera (receding along -z, light propagation on).  Its light sphere — an
    analytic object seen with aberration and light delay — is reproduced pixel for pixel (0 of the crop's 17 600 pixels differ) at
    the state in REFERENCE_SHOTS, and likewise at the fast end of the family of states the sphere cannot tell apart (0.95c after
    7.301 s: <= 8 pixels).  The bunny of this grab is NOT pinned: no camera state and no rotation of its `p` line reproduces its
    silhouette with the stand-in model (IoU 0.67), see tests/golden/make_reference_fixtures.py."""
    ref = _load("ref_mesh2_crop_y290_x1200.png")
    img = _render_top_down("mesh2", rows=(290, 400))[:, 1200:1360]
    assert np.abs(img - ref).max() == 0
    s = Scene.from_file("bunny")
    s.set_camera((0.0, 0.0, -math.tanh(9400 / 5000.0)), 7.301)
    s.update_objects()
    fast = _render_top_down("mesh2", rows=(290, 400), scene=s)[:, 1200:1360]
    assert (np.abs(fast - ref).max(axis=2) > 1).sum() <= 12
    rest = _render_top_down("mesh1", rows=(290, 400))[:, 1200:1360]                    # ... and it is not the resting camera's sphere
    assert (np.abs(rest - ref).max(axis=2) > 1).sum() > 200


def test_reference_screenshot_sphere_moving_boosted_textured_sphere():
    """Screenshots/sphere_moving.png (README.md:126-127, "Moving sphere"): the ball of sphere_stationary.png passing the
    resting camera with light propagation on: object boost, retarded position and the Terrell-rotated pattern of a
    TEXTURED sphere.  The grab was taken at 0.99c (not the 0.9c of the shipped Scenes/soccer.txt) with the ball
    0.99 x 4.5555 units along its path (REFERENCE_SHOTS): there 336 of the grab's 3.5 M pixels are more than 1 LSB off;
    0.4 ms, 0.0001c or 0.001 rad away it is more than ten thousand."""
    img = _render_top_down("sphere_moving")
    d = np.abs(img[::4, ::4] - _load("ref_sphere_moving_stride4.png")).max(axis=2)
    assert (d > 1).sum() <= 30, (d > 1).sum()
    ref_crop = _load("ref_sphere_moving_crop_y380_x960.png")
    here = (np.abs(img[380:1020, 960:1600] - ref_crop).max(axis=2) > 1).sum()
    assert here <= 400, here
    assert (np.abs(img - img[0, 0]).max(axis=2) > 8).sum() == 283634          # the ball's outline: the grab's own pixel count

    def off(text=None, dt=0.0):
        s = None
        if text:
            s = Scene()
            s.inputScene(text)
            s.set_camera((0, 0, 0), REFERENCE_SHOTS["sphere_moving"]["t"])
            s.update_objects()
        o = _render_top_down("sphere_moving", rows=(380, 1020), scene=s, dt=dt)[:, 960:1600]
        return (np.abs(o - ref_crop).max(axis=2) > 1).sum()
    text = REFERENCE_SHOTS["sphere_moving"]["text"]
    assert off(dt=0.0004) > 10000 and off(dt=-0.0004) > 10000
    assert off(text.replace("v0.99,", "v0.9901,")) > 8000
    assert off(text.replace("p0,0,5,2,", "p0,0,5,2.001,")) > 10000


def shadows_pear_mask():
    """Client-area pixels (top-down) whose primary ray hits the pear: the mesh object alone, light propagation off."""
    s = load_reference_shot("shadows1")
    pear = s.objects()[SHADOWS_PEAR_OBJECT:SHADOWS_PEAR_OBJECT + 1].copy()
    y0, y1, _, _ = SHADOWS_CROP
    px, _, _ = oracle_ffi.render(s, CLIENT_W, CLIENT_H, rows=(CLIENT_H - y1, CLIENT_H - y0), want_rgb=False, objects=pear, interval=0)
    img = px["rgba"].reshape(CLIENT_H, CLIENT_W, 4)[CLIENT_H - y1:CLIENT_H - y0][::-1, :, :3]
    return (img != img[0, 0]).any(axis=2)          # rows y0..y1 of the client area, every column


@pytest.mark.parametrize("shot,budget", [("shadows1", 32), ("shadows2", 40), ("shadows4", 40), ("shadows5", 64)])
def test_reference_screenshot_shadows_mesh_path(shot, budget):
    """Scenes/shadows.txt (README.md:117-122): the reference's own grabs of a scene with a MESH (the pear: OBJ import,
    smooth normals, octree build, octree walk of opencl_kernel.cl:200-308 for primary rays and for the shadow rays of
    every lit pixel), a light moving at 0.95c, light-delayed shadows.  Camera at rest; the clock (whole ms) recovered by
    tests/golden/fit_reference_camera.py.  At that clock the grab is reproduced to <= 1 LSB on all but a few dozen of
    3.5 M pixels — shadow-edge and silhouette pixels, where the author's GPU rounds differently (the reference builds
    with no options, so its contraction/division rounding is implementation-defined) — and on all but at most one of
    the pear's own 34 099 pixels; three milliseconds earlier or later several times as many pixels are wrong."""
    y0, y1, x0, x1 = SHADOWS_CROP
    img = _render_top_down(shot)
    ref4 = _load(f"ref_{shot}_stride4.png")
    d4 = np.abs(img[::4, ::4] - ref4).max(axis=2)
    assert (d4 > 1).sum() <= 8, f"stride-4 subsample: {(d4 > 1).sum()} pixels off by more than 1 level"
    assert (d4 > 0).mean() < 1e-3
    ref_crop = _load(f"ref_{shot}_crop_y{y0}_x{x0}.png")
    dc = np.abs(img[y0:y1, x0:x1] - ref_crop).max(axis=2)
    assert (dc > 1).sum() <= budget, f"full-resolution crop: {(dc > 1).sum()} pixels off by more than 1 level"
    assert (dc == 0).mean() > 0.998
    pear = shadows_pear_mask()[:, x0:x1]
    assert 33000 < pear.sum() < 35000                       # the pear's pixels: the mesh path proper
    assert (dc[pear] > 1).sum() <= 1 and (dc[pear] > 0).sum() <= 64, ((dc[pear] > 1).sum(), (dc[pear] > 0).sum())
    # the pin is sharp: 3 ms off, the light-delayed shadow edges and highlights have moved
    # (counted on the stride-4 subsample of the whole frame + the crop: the committed fixtures)
    def off_pixels(dt):
        o = _render_top_down(shot, dt=dt)
        return int((np.abs(o[::4, ::4] - ref4).max(axis=2) > 0).sum() + (np.abs(o[y0:y1, x0:x1] - ref_crop).max(axis=2) > 0).sum())
    here = off_pixels(0.0)
    assert off_pixels(-0.003) > 3 * here + 20 and off_pixels(0.003) > 3 * here + 20, (here, off_pixels(-0.003), off_pixels(0.003))


def _render_gif_sized(scene, t, v=(0, 0, 0)):
    """Oracle frame of a camera (at rest unless v is given) at clock t, box-filtered from the grab size down to the GIFs' 800x429."""
    scene.set_camera(v, t)
    scene.update_objects()
    px, _, _ = oracle_ffi.render(scene, CLIENT_W, CLIENT_H, want_rgb=False)
    img = px["rgba"].reshape(CLIENT_H, CLIENT_W, 4)[::-1, :, :3]
    return np.asarray(Image.fromarray(np.ascontiguousarray(img)).resize((800, 429), Image.BOX)).astype(np.int16)


def test_reference_gif_cubes_moving_objects_with_light_delay():
    """Screenshots/cubes.gif, frame 26: a line of cubes passing a resting camera at 0.9c, seen with light delay
    (object boosts + retarded positions), next to an identical line at rest.  Pinned to the resolution of a
    rescaled 256-colour image: silhouettes overlap to 98 %, colours inside agree to the palette step."""
    _, f, t = REFERENCE_GIF_FRAMES["cubes"][0]
    ref = _load(f"ref_cubes_gif_frame{f}.png")
    bg = np.array([47, 47, 76])
    scene = Scene.from_file("cubes")

    def compare(tt):
        img = _render_gif_sized(scene, tt)
        a, b = np.abs(img - bg).max(axis=2) > 12, np.abs(ref - bg).max(axis=2) > 12
        return (a & b).sum() / (a | b).sum(), np.abs(img - ref)[a & b].mean()
    iou, colour = compare(t)
    assert iou > 0.975 and colour < 10.0, (iou, colour)
    iou_off, _ = compare(t + 0.35)          # the moving line half a cube spacing further on
    assert iou_off < 0.93, iou_off


def test_reference_gif_ladder_paradox_moving_objects():
    """Screenshots/ladder_paradox_garage_frame.gif: ruler and garage doors at 0.9c (x and y), light propagation
    off.  Three frames 40 GIF frames apart fit camera clocks 1.56 s = 40 x 39 ms apart — the GIF's own 40 ms
    frame time — and each matches visibly better there than 0.2 s earlier or later."""
    scene = Scene.from_file("ladder_paradox")
    crop = (slice(150, 270), slice(380, 700))        # the garage: everything that moves
    for _, f, t in REFERENCE_GIF_FRAMES["ladder"]:
        ref = _load(f"ref_ladder_gif_frame{f}.png")[crop]
        err = {dt: np.abs(_render_gif_sized(scene, t + dt)[crop] - ref).mean() for dt in (-0.2, 0.0, 0.2)}
        assert err[0.0] < 4.5, (f, err)
        assert err[-0.2] > 1.15 * err[0.0] and err[0.2] > 1.15 * err[0.0], (f, err)


def test_reference_gif_ladder_paradox_from_the_moving_ladder():
    """Screenshots/ladder_paradox_ladder_frame.gif: the same scene seen from a camera that moves with the ladder at 0.9c
    (README.md:105-107), light propagation off: a MOVING CAMERA and MOVING OBJECTS together (the garage's ruler and walls pass
    at 0.9c, the doors move diagonally and come out slanted: relativity of simultaneity).  At v = tanh(7361/5000) c three
    frames 40 GIF frames apart fit clocks 1.575 s apart — 39.4 ms per GIF frame, the file's frame time — and each matches
    better there than 0.1 s earlier or later, and by more than 12% than 0.4 s away."""
    from conftest import LADDER_FRAME_CAMERA_V
    scene = Scene.from_file("ladder_paradox")
    crop = (slice(150, 270), slice(60, 760))         # the corridor: both rulers and the doors
    for _, f, t in REFERENCE_GIF_FRAMES["ladderframe"]:
        ref = _load(f"ref_ladderframe_gif_frame{f}.png")[crop]
        err = {dt: np.abs(_render_gif_sized(scene, t + dt, LADDER_FRAME_CAMERA_V)[crop] - ref).mean() for dt in (-0.4, -0.1, 0.0, 0.1, 0.4)}
        assert err[0.0] < 3.5, (f, err)
        assert err[-0.1] > err[0.0] and err[0.1] > err[0.0], (f, err)
        assert err[-0.4] > 1.12 * err[0.0] and err[0.4] > 1.12 * err[0.0], (f, err)


# SURVEY.md §8(a) "Per-primary-ray work [probe, 1920x1080]" — measured from the reference itself
SURVEY_PER_RAY = {
    "bunny": dict(shadow_rays=0.027, sphere_tests=1.00, cube_tests=0.0, octree_calls=1.027, root_aabb_hits=0.119,
                  leaf_visits=0.671, descent_steps=0.694, inside_descent_steps=0.147, tri_tests=1.188, pixels_hit=0.042),
    "shadows": dict(shadow_rays=0.551, sphere_tests=2.55, cube_tests=3.07, octree_calls=1.514, root_aabb_hits=0.228,
                    leaf_visits=1.157, descent_steps=1.358, tri_tests=3.568, pixels_hit=0.562),
    "arch": dict(shadow_rays=0.296, sphere_tests=1.00, cube_tests=5.18, octree_calls=0.0, root_aabb_hits=0.0,
                 leaf_visits=0.0, descent_steps=0.0, tri_tests=0.0, pixels_hit=0.296),
}


@pytest.mark.parametrize("name", list(SURVEY_PER_RAY))
def test_per_ray_work_matches_survey_probe(name):
    scene = load_config(name)
    W, H = 1920, 1080
    _, _, st = oracle_ffi.render(scene, W, H, want_rgb=False, want_stats=True)
    for key, want in SURVEY_PER_RAY[name].items():
        got = st[key] / (W * H)
        digits = 2 if want in (2.55, 3.07, 5.18, 1.00) else 3      # as many digits as the survey printed
        # one unit in the survey's last printed digit (its host matrices came from a g++ build whose
        # sqrt/sin/cos resolved to double, a <= 1 ulp difference in Object[] that moves a few hundred tests)
        assert abs(got - want) <= 10 ** -digits, f"{name}.{key}: {got} vs survey {want}"


@pytest.mark.parametrize("name", list(CONFIGS))
def test_oracle_golden_frames(name):
    g = np.load(os.path.join(GOLDEN, f"oracle_{name}_128x72.npz"))
    scene = load_config(name)
    assert np.array_equal(scene.buffers()["objects"], g["objects"]), "Object[] bytes drifted"
    px, rgb, _ = oracle_ffi.render(scene, 128, 72, threads=3)
    assert np.array_equal(px["rgba"].reshape(72, 128, 4), g["rgba"])
    assert np.array_equal(rgb.view(np.uint32), g["rgb"].view(np.uint32))


def test_oracle_threads_and_row_ranges_agree():
    scene = load_config("shadows")
    full, rgb_full, _ = oracle_ffi.render(scene, 200, 120, threads=1)
    part, rgb_part, _ = oracle_ffi.render(scene, 200, 120, rows=(40, 77), threads=5)
    sl = slice(40 * 200, 77 * 200)
    assert np.array_equal(part["rgba"][sl], full["rgba"][sl]) and np.array_equal(rgb_part[40:77], rgb_full[40:77])
    assert not part["rgba"][:40 * 200].any() and not part["rgba"][77 * 200:].any()


def test_pixel_record_layout():
    scene = load_config("cube")
    px, _, _ = oracle_ffi.render(scene, 64, 48)
    assert px.dtype.itemsize == 16
    assert np.array_equal(px["x"], np.tile(np.arange(64, dtype=np.float32), 48))
    assert np.array_equal(px["y"], np.repeat(np.arange(48, dtype=np.float32), 64))
    assert np.all(px["rgba"][:, 3] == 1)
    # background = Hable((0.15,0.15,0.25))/Hable(1) packed: (47,47,76) as in the reference's screenshots
    assert tuple(px["rgba"][0][:3]) == (47, 47, 76)


def test_oracle_asin_atan2_accuracy():
    """The oracle's explicit asin / atan2 (OpenCL leaves their last bits to the implementation): asin within 1 ulp and
    atan2 within 2 ulp of the exact value on dense and edge inputs, exact special cases, odd symmetry."""
    import ctypes as C
    lib = oracle_ffi.lib()
    out = (C.c_float * 2)()

    def both(a, y, x):
        lib.rpt_oracle_asin_atan2(float(a), float(y), float(x), out)
        return np.float32(out[0]), np.float32(out[1])

    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(-1, 1, 40000), np.linspace(-1, 1, 4001), [0.0, 0.5, -0.5, 0.4999999, 2.0 ** -13, 1.0, -1.0]]).astype(np.float32)
    yx = rng.normal(size=(a.size, 2)).astype(np.float32) * rng.choice([1e-6, 1e-2, 1.0, 1e3], size=(a.size, 1)).astype(np.float32)
    yx[:8] = [[0, 1], [0, -1], [1, 0], [-1, 0], [1, 1], [-1, -1], [1e-30, 1e30], [1e30, -1e-30]]
    worst_s = worst_t = 0.0
    for k in range(a.size):
        s, t = both(a[k], yx[k, 0], yx[k, 1])
        ws, wt = np.arcsin(np.float64(a[k])), np.arctan2(np.float64(yx[k, 0]), np.float64(yx[k, 1]))
        worst_s = max(worst_s, abs(np.float64(s) - ws) / np.spacing(np.float32(abs(ws)) if ws else np.float32(1e-30)))
        worst_t = max(worst_t, abs(np.float64(t) - wt) / np.spacing(np.float32(abs(wt)) if wt else np.float32(1e-30)))
    assert worst_s < 1.0 and worst_t < 2.0, (worst_s, worst_t)       # OpenCL allows 4 and 6 ulp
    assert both(0.25, 0.3, -0.7)[0] == -both(-0.25, 0.3, -0.7)[0]
    assert both(0, 0.3, -0.7)[1] == -both(0, -0.3, -0.7)[1]
    assert np.isnan(both(1.5, 1, 1)[0]) and np.isnan(both(0, np.nan, 1)[1])
    assert both(0, 0.0, 1.0)[1] == 0 and both(0, 0.0, -1.0)[1] == np.float32(np.pi)


def test_msaa_one_is_the_shipped_path_and_more_samples_only_touch_edges():
    """rpt_oracle_args.msaa: 0 and 1 are the reference as shipped (MSAASAMPLES = 1, opencl_kernel.cl:7) — the path the golden
    screenshots pin; 2 runs the loop of :641-648.  Where all four samples of a pixel see the same thing nothing changes beyond the
    rounding of (c + c + c + c) / 4; pixels that change by more lie on silhouettes and shadow borders."""
    from relativitypathtracer_amd import Scene
    scene = Scene.from_file("bunny")      # flat background, one shaded mesh, one light sphere
    scene.update_objects()
    W, H = 240, 135
    p0, _, _ = oracle_ffi.render(scene, W, H, want_rgb=False)
    p1, _, _ = oracle_ffi.render(scene, W, H, want_rgb=False, msaa=1)
    p2, _, _ = oracle_ffi.render(scene, W, H, want_rgb=False, msaa=2)
    assert np.array_equal(p0["rgba"], p1["rgba"])
    d = np.abs(p2["rgba"][:, :3].astype(np.int16) - p0["rgba"][:, :3].astype(np.int16)).max(axis=1)
    changed = int((d > 1).sum())
    assert 0 < changed < 0.05 * W * H, changed
    sky = (p0["rgba"][:, :3] == p0["rgba"][0, :3]).all(axis=1).reshape(H, W)          # the background colour of pixel (0, 0)
    inner = sky[1:-1, 1:-1] & sky[:-2, 1:-1] & sky[2:, 1:-1] & sky[1:-1, :-2] & sky[1:-1, 2:]
    assert inner.sum() > 100 and d.reshape(H, W)[1:-1, 1:-1][inner].max() <= 1
