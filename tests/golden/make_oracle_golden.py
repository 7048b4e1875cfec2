#!/usr/bin/env python3
"""Golden frames of the CPU oracle (NOT of the reference — see DESIGN.md §3 for why the reference
itself cannot be built here): every benchmark/test configuration at 128x72, packed bytes and float RGB.
They freeze the oracle's output so that (a) oracle drift is caught on CPU and (b) the HIP path is
compared with a committed record on the GPU box as well as with the live oracle.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_ffi                      # noqa: E402
from conftest import CONFIGS, load_config   # noqa: E402

W, H = 128, 72
for name in CONFIGS:
    scene = load_config(name)
    px, rgb, _ = oracle_ffi.render(scene, W, H, threads=1)
    np.savez_compressed(os.path.join(HERE, f"oracle_{name}_{W}x{H}.npz"), rgba=px["rgba"].reshape(H, W, 4), rgb=rgb,
                        objects=scene.buffers()["objects"])
    print(name, "hit pixels", int((rgb.reshape(-1, 3) != rgb.reshape(-1, 3)[0]).any(axis=1).sum()))
