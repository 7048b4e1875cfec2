#!/usr/bin/env python3
"""Cuts test fixtures out of the reference's own output images (its README screenshots).

Run in the build container only (needs /root/reference).  The fixtures are DATA the reference
ships — window grabs of its own renderer — not source: `Screenshots/cube1.png` (Scenes/cube.txt,
stationary camera) and `Screenshots/arch1.png` (Scenes/arch.txt, stationary camera).  Both scenes
are static (every velocity 0), so the image does not depend on the unrecorded camera time and is
reproducible from the scene file alone.  The 2560x1400 grabs carry a 23-row title bar; the client
area is 2560x1377.

Written: an exact stride-4 subsample of each client area (every 4th pixel of every 4th row, no
filtering) and one full-resolution 640x400 crop of arch1 around the arch, light and shadows.
"""
import os
import numpy as np
from PIL import Image

SRC = "/root/reference/Screenshots"
DST = os.path.dirname(os.path.abspath(__file__))
TITLE_BAR = 23

for name in ("cube1", "arch1"):
    im = np.asarray(Image.open(os.path.join(SRC, name + ".png")).convert("RGB"))[TITLE_BAR:]
    assert im.shape == (1377, 2560, 3), im.shape
    Image.fromarray(np.ascontiguousarray(im[::4, ::4])).save(os.path.join(DST, f"ref_{name}_stride4.png"), optimize=True)
    if name == "arch1":
        Image.fromarray(np.ascontiguousarray(im[300:700, 960:1600])).save(os.path.join(DST, "ref_arch1_crop_y300_x960.png"), optimize=True)
print("ok")
