"""The C-ABI libraries load and export every symbol their headers declare (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(rpt_[a-z0-9_]+)\s*\(", text))
    return {n for n in names if not n.endswith("_decoder")}   # function-pointer typedef


def test_hip_library_exports_every_declared_symbol():
    from relativitypathtracer_amd import _ffi
    lib = C.CDLL(_ffi.hip_lib_path())
    names = declared("rpt.h")
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), f"librpt_hip.so does not export {n}"
    assert names == set(_ffi.HIP_SYMBOLS), (names ^ set(_ffi.HIP_SYMBOLS))
    assert b"gfx950" in _ffi.hip().rpt_version()   # binds all argtypes; rpt_version needs no device


def test_scene_library_exports_every_declared_symbol():
    from relativitypathtracer_amd import _ffi
    lib = _ffi.scene_lib()
    for n in sorted(declared("rpt_scene.h")):
        assert hasattr(lib, n), f"librpt_scene.so does not export {n}"


def test_layout_sizes_match_reference():
    from relativitypathtracer_amd import _ffi
    from relativitypathtracer_amd.scene import OBJECT_DTYPE, OCTREE_DTYPE
    assert C.sizeof(_ffi.Object) == 320 and OBJECT_DTYPE.itemsize == 320      # Object.h:6-22, SURVEY App. D
    assert C.sizeof(_ffi.Octree) == 96 and OCTREE_DTYPE.itemsize == 96         # Octree.h:4-12
    offs = {n: getattr(_ffi.Object, n).offset for n, *_ in _ffi.Object._fields_}
    assert (offs["InvM"], offs["Lorentz"], offs["InvLorentz"], offs["stationaryCam"], offs["color"]) == (64, 128, 192, 256, 272)
    assert (offs["type"], offs["meshIndex"], offs["textureIndex"], offs["textureWidth"], offs["textureHeight"]) == (288, 292, 296, 300, 304)
    assert (offs["light"], offs["flashPeriod"], offs["flashDuration"]) == (308, 312, 316)


def test_no_cpu_fallback_without_gpu():
    """Without a device the product path fails loudly instead of silently rendering on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from relativitypathtracer_amd.renderer import Renderer, RenderError
    with pytest.raises(RenderError):
        Renderer(0)


def test_product_does_not_touch_the_oracle():
    """Nothing under the package or include/ references oracle/ (the oracle is test infrastructure)."""
    for base in ("relativitypathtracer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".cpp", ".hip", ".hpp")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "librpt_oracle" not in text and "oracle_ffi" not in text and "rpt_oracle.h" not in text, os.path.join(dirpath, f)
