"""The measurement arms of rounds 1-3 (librpt_hip_diag.so: csrc/rpt_diag_kernels.hip.h) against the CPU oracle.

None of these kernels is the product — each lost an A/B (profiles/r03_*.txt) or is an older default kept as the
baseline — but every one of them claims to compute exactly what the product computes, and a measurement taken on a kernel that
renders something else would be worthless.  So they are held to the product's bar: float RGB bit-identical to the oracle.
  26        round 1: per-tile object masks from a prepass kernel
  40, 42    the default at 4 / 6 waves per SIMD
  141, 143  round 2's octree walk (node-policy form), natural order / mesh band first
  60, 62, 63  persistent workgroups: LDS-staged rectangles + octree top + triangle records / the same with triangle records
              from global memory / the persistent skeleton around the per-pixel trace
  61        the per-workgroup LDS ray queue
  256+f     the product walk with its round trips re-ordered further (flags: rpt_diag_walks.hip.h)
  529, 541, 621   the six neighbour indices read with the node record (/ + records ahead / + the first triangle too: a step in one round trip)
  561, 573  = kernel 43's latency walk, natural order / mesh band first;  2573 the same at four waves per SIMD
  589       the latency walk with the packed leaf count;  625, 637 kernel 41's / 43's walk with the triangle id read with every record
  593, 605  the 16^3 root table (descend_from_root) in kernel 41's / 43's walk
  641, 653  the lanes of a wave along the Z curve through its tile
  657, 669, 673   records read once per wave where the wave stands in one node: through the scalar cache / by one lane + readfirstlane
  705, 717  round 4: kernel 41's / 43's walk WITHOUT the repeated triangle tests (list entries whose triangle was in the previous
            leaf's list are skipped: exact, see octree_walk's DEDUP in csrc/rpt_kernels.hip.h)
"""
import numpy as np
import pytest

import oracle_ffi
from conftest import load_config

pytestmark = pytest.mark.gpu

ARMS = [26, 40, 42, 141, 143, 60, 61, 62, 63, 256, 257, 259, 261, 263, 265, 269, 273, 277, 285, 305, 317, 337, 349, 401, 1257, 2257, 2259, 2263, 529, 541, 561, 573, 589, 593, 605, 621, 625, 637, 641, 653, 657, 669, 673, 689, 701, 2573, 705, 717]
SCENES = {"bunny": (480, 270), "shadows": (480, 270), "arch": (480, 270), "cubes": (320, 184), "soccer": (320, 184), "cube": (333, 77)}


@pytest.fixture(scope="module")
def diag_renderer():
    from relativitypathtracer_amd.renderer import Renderer
    r = Renderer(0, diag=True)
    yield r
    r.close()


@pytest.fixture(scope="module")
def oracle_frames():
    out = {}
    for name, (W, H) in SCENES.items():
        scene = load_config(name)
        out[name] = (scene,) + oracle_ffi.render(scene, W, H)[:2]
    return out


@pytest.mark.parametrize("variant", ARMS)
def test_arm_matches_oracle(diag_renderer, oracle_frames, variant):
    r = diag_renderer
    for name, (W, H) in SCENES.items():
        scene, opx, orgb = oracle_frames[name]
        r.set_variant(variant)
        r.upload_scene(scene)
        r.set_scene_params(scene, W, H)
        r.set_rows(0, 1, False)
        r.set_output(None)
        r.set_debug_rgb(True)
        for blocking in (True, False):
            if blocking:
                r.render()
            else:
                r.render_async()
                r.sync()
            px, rgb = r.read_framebuffer(), r.read_debug_rgb()
            assert np.array_equal(rgb.view(np.uint32), orgb.view(np.uint32)), f"variant {variant} {name}: float RGB not bit-identical"
            assert np.array_equal(px["rgba"], opx["rgba"]), f"variant {variant} {name}: packed bytes differ"


def test_persistent_kernels_twice_in_a_row(diag_renderer, oracle_frames):
    """The persistent kernels count their claims in one of two counter sets and zero the other for the next launch: frames
    rendered back to back (no host wait between them), with a resolution change in between, must all be complete."""
    r = diag_renderer
    scene, opx, orgb = oracle_frames["bunny"]
    W, H = SCENES["bunny"]
    for variant in (60, 62, 63):
        r.set_variant(variant)
        r.upload_scene(scene)
        r.set_scene_params(scene, W, H)
        r.set_output(None)
        r.set_debug_rgb(False)
        for _ in range(5):
            r.render_async()
        r.sync()
        assert np.array_equal(r.read_framebuffer()["rgba"], opx["rgba"]), variant
        r.set_scene_params(scene, 640, 360)
        r.render()
        r.set_scene_params(scene, W, H)
        r.render()
        assert np.array_equal(r.read_framebuffer()["rgba"], opx["rgba"]), variant
