"""The opt-in "OpenCL-conformant arithmetic" build of the kernel (variants 50 / 51: fma contraction, 2.5-ulp division, 3-ulp
sqrt — csrc/rpt_relaxed.hip).  It is NOT the parity path and not bit-exact; this file pins how far it is from the oracle so
that any number measured with it can be read in context: per configuration the largest |dRGB| on the tonemapped floats,
the number of pixels beyond the 1e-4 tolerance, and the number whose packed bytes differ by more than one level (silhouette
and shadow-edge pixels, where a different rounding picks a different hit) — the kind and size of difference the reference's
own GPU shows against the exact oracle (218 of 3.5 M pixels of Screenshots/shadows4.png)."""
import json
import os

import numpy as np
import pytest

import oracle_ffi
from conftest import CONFIGS, load_config

pytestmark = pytest.mark.gpu
SIZES = {"cube": (640, 480), "arch": (960, 540), "bunny": (960, 540), "shadows": (960, 540), "cubes": (960, 540), "soccer": (640, 360)}


@pytest.fixture(scope="module")
def renderer():
    from relativitypathtracer_amd.renderer import Renderer
    r = Renderer(0)
    yield r
    r.close()


@pytest.mark.parametrize("name", list(SIZES))
def test_relaxed_arithmetic_stays_close_but_is_not_the_parity_path(renderer, name, record_property):
    W, H = SIZES[name]
    scene = load_config(name)
    opx, orgb, _ = oracle_ffi.render(scene, W, H)
    out = {}
    for variant in (0, 50, 51):
        renderer.set_variant(variant)
        renderer.upload_scene(scene)
        renderer.set_scene_params(scene, W, H)
        renderer.set_rows(0, 1, False)
        renderer.set_output(None)
        renderer.set_debug_rgb(True)
        renderer.render()
        px, rgb = renderer.read_framebuffer(), renderer.read_debug_rgb()
        d = np.abs(rgb - orgb).max(axis=2)
        dbytes = np.abs(px["rgba"].astype(np.int16) - opx["rgba"].astype(np.int16)).max(axis=1)
        out[variant] = {"max_abs_drgb": float(d.max()), "pixels_beyond_1e-4": int((d > 1e-4).sum()), "pixels_beyond_1_level": int((dbytes > 1).sum()),
                        "pixels": W * H}
    renderer.set_variant(0)
    record_property("relaxed_vs_oracle", json.dumps(out))
    print(name, json.dumps(out))
    assert out[0]["max_abs_drgb"] == 0.0 and out[0]["pixels_beyond_1_level"] == 0           # the default stays exact
    for v in (50, 51):
        assert out[v]["pixels_beyond_1e-4"] < 0.005 * W * H, (name, v, out[v])            # a fraction of a percent: edges only
        assert out[v]["pixels_beyond_1_level"] < 0.003 * W * H, (name, v, out[v])
