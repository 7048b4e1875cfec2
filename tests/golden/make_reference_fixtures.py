#!/usr/bin/env python3
"""Cuts test fixtures out of the reference's own output images (its README screenshots).

Run in the build container only (needs /root/reference).  The fixtures are DATA the reference
ships — window grabs of its own renderer — not source.  The 2560x1400 grabs carry a 23-row title
bar; the client area is 2560x1377.

  cube1.png, arch1.png   Scenes/cube.txt, Scenes/arch.txt with the camera at rest.  Both scenes are static
                         (every velocity 0), so the image does not depend on the unrecorded camera time
                         and is reproducible from the scene file alone.
  cube2.png, cube3.png   Scenes/cube.txt "moving frame (0.9c to the right)" without / with light propagation
  arch2.png              Scenes/arch.txt "camera moving towards the arch at 0.95c" (README.md:81-94).
                         The exact velocity and clock of these grabs are not recorded; they were recovered
                         by tests/golden/fit_reference_camera.py (see there) and are kept in
                         tests/conftest.py::REFERENCE_SHOTS.

  cubes.gif, ladder_paradox_garage_frame.gif
                         800x429 palette animations of Scenes/cubes.txt (a line of cubes passing at 0.9c, light
                         propagation on) and Scenes/ladder_paradox.txt (ruler and garage doors at 0.9c, light
                         propagation off) from a camera at rest: single frames, converted to RGB as they are.  They
                         pin the geometry of MOVING OBJECTS to the resolution of a rescaled palette image; their clock
                         values (tests/conftest.py::REFERENCE_GIF_FRAMES) were found like the grabs' — and come out
                         40 ms per frame apart for the ladder, the GIF's own frame time.
  ladder_paradox_ladder_frame.gif
                         the same scene from a camera moving WITH the ladder (0.9c, light propagation off): a moving camera and
                         moving objects together, the door slanted by the relativity of simultaneity.  Velocity tanh(7361/5000) c
                         (the rapidity step nearest 0.9c), clocks 3.075 / 4.65 / 6.225 s for frames 60 / 100 / 140: 39.4 ms apart
                         per GIF frame again.

  shadows1/2/4/5.png     Scenes/shadows.txt (README.md:117-122) from a camera at rest with light propagation on: a light
                         sphere crossing the scene at 0.95c, two cubes, a sphere and the pear MESH (Models/pear.obj through
                         the OBJ loader, the octree builder and intersect_octree / intersect_triangle / intersect_AABB /
                         getOppositeBoxSide — opencl_kernel.cl:106-308 — for primary AND shadow rays).  The only unknown of
                         these grabs is the camera clock (whole milliseconds); tests/golden/fit_reference_camera.py
                         finds 6.157 s, 9.212 s, 18.229 s and 25.987 s.  The crop holds the pear, the part of its shadow
                         next to it and (shadows4/5) the light.

  sphere_stationary.png  Scenes/soccer.txt "Stationary sphere" (README.md:124-125): a textured sphere at rest, turned by
                         2 rad about y when the grab was taken (recovered by fit_reference_camera.py): sphere (u,v) through
                         atan2/asin and the bilinear fetch.
  sphere_moving.png      "Moving sphere" (README.md:126-127): the same turned ball passing the resting camera with light
                         propagation on — at 0.99c, not the 0.9c of the shipped file (fit_reference_camera.py::
                         fit_moving_sphere): the boost of a textured sphere, its retarded position and Terrell-rotated
                         pattern.  With `v0.99,0,0` and the ball 0.99 x 4.5555 units along its path, 336 of the grab's
                         3 525 120 pixels are more than 1 LSB off.

  mesh1.png              Scenes/bunny.txt — THE HEADLINE SCENE — from a camera at rest (README.md:85-87).  The grab shows
                         Models/StanfordBunny.obj, which the reference tree lacks (.MISSING_LARGE_BLOBS); the stand-in
                         Models/bunny.obj is the same model in the same pose under ANOTHER NORMALISATION (its silhouette is 0.78 x
                         the grab's and shifted), so the mesh pixels cannot be pinned.  What the grab does pin for this scene:
                         the light sphere (an analytic object: its 1 009 pixels, every one of them, byte for byte), the
                         background, the camera's framing and the `p`/`c`/`l`/`A` lines of the scene file; and, up to a
                         similarity of the image plane, the bunny's pose (silhouette IoU 0.94 at scale 1.28).  The crop holds
                         the light sphere.

  mesh2.png, mesh3.png   two more grabs of the bunny scene that the README does not describe.  mesh2: the light sphere sits 5 pixels higher
                         than in mesh1 and the bunny shows another side.  The sphere is reproduced EXACTLY (0 of its 1 079 pixels off) by
                         a camera receding along -z with light propagation on — by a whole one-parameter family of such states (from
                         0.04c after 3.07 s to 0.95c after 7.30 s: the analytic sphere cannot tell them apart) — and the bunny's silhouette
                         by none of them (IoU 0.67 under the mesh1 similarity, against 0.94 for mesh1), nor by any rotation of the `p` line
                         about ten axes: the grab's exact scene line is not recoverable, and the model file is missing anyway.  Kept: the
                         light sphere's crop, as a second exact pin of an analytic object seen from a MOVING camera with light delay.
                         mesh3: the camera stands next to / inside the bunny (primary rays starting inside the mesh's root box,
                         opencl_kernel.cl:233-248), no analytic object in view, a model the tree lacks: nothing of it can be pinned; the
                         regime is tested HIP-against-oracle (tests/test_gpu_parity.py::test_camera_inside_the_mesh_root_box).
                         (Found by /tmp fits during round 4; the numbers are in DESIGN.md section 3.)

Written per image: an exact stride-4 subsample of the client area (every 4th pixel of every 4th row, no
filtering) and one full-resolution crop of the part with the most detail.
"""
import os
import numpy as np
from PIL import Image

SRC = "/root/reference/Screenshots"
DST = os.path.dirname(os.path.abspath(__file__))
TITLE_BAR = 23

# name -> full-resolution crop (y0, y1, x0, x1) in client-area coordinates, or None
CROPS = {
    "cube1": None,
    "arch1": (300, 700, 960, 1600),      # arch, light, shadows
    "cube2": (826, 1377, 1150, 1410),    # the length-contracted crate
    "cube3": (826, 1377, 1000, 1620),    # the crate as seen with light delay (Terrell rotation)
    "arch2": (900, 1300, 960, 1600),     # brick floor under the arch: the most position-sensitive texture
    "sphere_stationary": (380, 1020, 960, 1600),   # the whole ball
    "sphere_moving": (380, 1020, 960, 1600),       # the whole ball, seen at 0.99c with light delay
    "shadows1": (540, 980, 1200, 1720),  # the pear (mesh path), dimly lit from the left
    "shadows2": (540, 980, 1200, 1720),
    "shadows4": (540, 980, 1200, 1720),  # the pear lit from the right, its shadow on the wall, the light sphere
    "shadows5": (540, 980, 1200, 1720),
    "mesh1": (300, 390, 1230, 1330),     # the light sphere of Scenes/bunny.txt
}

for name, crop in CROPS.items():
    im = np.asarray(Image.open(os.path.join(SRC, name + ".png")).convert("RGB"))[TITLE_BAR:]
    assert im.shape == (1377, 2560, 3), im.shape
    Image.fromarray(np.ascontiguousarray(im[::4, ::4])).save(os.path.join(DST, f"ref_{name}_stride4.png"), optimize=True)
    if crop:
        y0, y1, x0, x1 = crop
        Image.fromarray(np.ascontiguousarray(im[y0:y1, x0:x1])).save(os.path.join(DST, f"ref_{name}_crop_y{y0}_x{x0}.png"), optimize=True)
# mesh2.png: the light sphere only (full resolution), see above
im = np.asarray(Image.open(os.path.join(SRC, "mesh2.png")).convert("RGB"))[TITLE_BAR:]
Image.fromarray(np.ascontiguousarray(im[290:400, 1200:1360])).save(os.path.join(DST, "ref_mesh2_crop_y290_x1200.png"), optimize=True)
for gif, frames, stem in (("cubes", (26,), "cubes"), ("ladder_paradox_garage_frame", (60, 100, 140), "ladder"),
                         ("ladder_paradox_ladder_frame", (60, 100, 140), "ladderframe")):
    im = Image.open(os.path.join(SRC, gif + ".gif"))
    for f in frames:
        im.seek(f)
        im.convert("RGB").save(os.path.join(DST, f"ref_{stem}_gif_frame{f}.png"), optimize=True)
print("ok")
