"""ctypes bindings of the two in-tree shared libraries.

* ``librpt_scene.so`` — host-only scene front-end (``include/rpt_scene.h``)
* ``librpt_hip.so``   — the HIP render path for gfx950 (``include/rpt.h``)

The libraries are looked up next to this file (they are built in-tree by
``relativitypathtracer_amd/csrc/Makefile`` / ``__graft_entry__.build()``).  There is no CPU
fallback for the render path: if ``librpt_hip.so`` is missing or fails to load, ``hip()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class Float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Object(C.Structure):
    """rpt_object — 320 B (reference Object.h:6-22)."""
    _fields_ = [
        ("M", Float4 * 4), ("InvM", Float4 * 4), ("Lorentz", Float4 * 4), ("InvLorentz", Float4 * 4),
        ("stationaryCam", Float4), ("color", Float4),
        ("type", C.c_int32), ("meshIndex", C.c_int32), ("textureIndex", C.c_int32),
        ("textureWidth", C.c_int32), ("textureHeight", C.c_int32),
        ("light", C.c_uint8), ("_pad", C.c_uint8 * 3),
        ("flashPeriod", C.c_float), ("flashDuration", C.c_float),
    ]


class Octree(C.Structure):
    """rpt_octree — 96 B (reference Octree.h:4-12)."""
    _fields_ = [
        ("min", Float4), ("max", Float4), ("trisIndex", C.c_int32), ("trisCount", C.c_int32),
        ("children", C.c_int32 * 8), ("neighbors", C.c_int32 * 6),
    ]


class SceneDesc(C.Structure):
    """rpt_scene_desc — the eight scene arrays as {pointer, count} pairs."""
    _fields_ = [
        ("objects", C.c_void_p), ("object_count", C.c_size_t),
        ("vertices", C.c_void_p), ("vertex_count", C.c_size_t),
        ("normals", C.c_void_p), ("normal_count", C.c_size_t),
        ("uvs", C.c_void_p), ("uv_count", C.c_size_t),
        ("triangles", C.c_void_p), ("triangle_words", C.c_size_t),
        ("octrees", C.c_void_p), ("octree_count", C.c_size_t),
        ("octreeTris", C.c_void_p), ("octree_tri_count", C.c_size_t),
        ("textures", C.c_void_p), ("texture_bytes", C.c_size_t),
    ]


assert C.sizeof(Object) == 320 and C.sizeof(Octree) == 96 and C.sizeof(SceneDesc) == 128

TextureDecoder = C.CFUNCTYPE(C.c_int, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                             C.POINTER(C.c_int), C.c_void_p)

_scene_lib = None
_hip_lib = None


def _path(name: str) -> str:
    return os.path.join(_HERE, name)


def scene_lib() -> C.CDLL:
    global _scene_lib
    if _scene_lib is None:
        p = _path("librpt_scene.so")
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C relativitypathtracer_amd/csrc`")
        lib = C.CDLL(p)
        P, I, S = C.c_void_p, C.c_int, C.c_char_p
        FP = C.POINTER(C.c_float)
        sig = {
            "rpt_scene_create": (P, []),
            "rpt_scene_destroy": (None, [P]),
            "rpt_scene_last_error": (S, [P]),
            "rpt_scene_set_asset_root": (I, [P, S]),
            "rpt_scene_add_alias": (I, [P, S, S]),
            "rpt_scene_set_texture_decoder": (I, [P, TextureDecoder, P]),
            "rpt_scene_input": (I, [P, S]),
            "rpt_scene_read_obj": (I, [P, S]),
            "rpt_scene_read_obj_geometry": (I, [P, S, C.POINTER(C.c_size_t)]),
            "rpt_scene_append_octree": (I, [P, P, C.c_size_t, P, C.c_size_t]),
            "rpt_scene_read_texture": (I, [P, S]),
            "rpt_scene_add_texture_rgb8": (I, [P, P, I, I]),
            "rpt_scene_set_camera": (I, [P, FP, FP]),
            "rpt_scene_get_camera": (I, [P, FP, FP]),
            "rpt_scene_accelerate": (I, [P, FP, I]),
            "rpt_scene_reset_velocity": (I, [P]),
            "rpt_scene_set_paused": (I, [P, I]),
            "rpt_scene_advance_time": (I, [P, I]),
            "rpt_scene_set_interval": (I, [P, I]),
            "rpt_scene_toggle_interval": (I, [P]),
            "rpt_scene_update_objects": (I, [P]),
            "rpt_scene_get_desc": (I, [P, C.POINTER(SceneDesc)]),
            "rpt_scene_get_params": (I, [P, FP, FP, C.POINTER(C.c_int)]),
            "rpt_scene_get_velocities": (I, [P, C.POINTER(P), C.POINTER(C.c_size_t)]),
            "rpt_scene_get_mesh_roots": (I, [P, C.POINTER(P), C.POINTER(C.c_size_t)]),
            "rpt_write_ppm": (I, [S, P, I, I]),
            "rpt_write_png": (I, [S, P, I, I]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _scene_lib = lib
    return _scene_lib


HIP_SYMBOLS = {
    # name: (restype, argtypes) — exactly the entry points include/rpt.h declares
    "rpt_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "rpt_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]),
    "rpt_destroy": (None, [C.c_void_p]),
    "rpt_last_error": (C.c_char_p, [C.c_void_p]),
    "rpt_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "rpt_share_scene": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpt_set_objects": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rpt_set_params": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, C.c_int]),
    "rpt_set_output": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpt_set_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "rpt_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpt_set_debug_rgb": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpt_set_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "rpt_object_screen_rect": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rpt_object_screen_bounds": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rpt_mesh_segment_cull_record": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "rpt_object_screen_bounds_proposed": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rpt_certify_screen_bounds": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "rpt_verify_frame": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rpt_last_variant": (C.c_int, [C.c_void_p]),
    "rpt_set_msaa": (C.c_int, [C.c_void_p, C.c_int]),
    "rpt_probe_walk": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "rpt_timing_end_spans": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "rpt_probe_division": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_void_p, C.c_int]),
    "rpt_probe_object": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "rpt_render": (C.c_int, [C.c_void_p]),
    "rpt_render_async": (C.c_int, [C.c_void_p]),
    "rpt_sync": (C.c_int, [C.c_void_p]),
    "rpt_output_ptr": (C.c_void_p, [C.c_void_p]),
    "rpt_output_bytes": (C.c_size_t, [C.c_void_p]),
    "rpt_read_framebuffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rpt_read_debug_rgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rpt_last_frame_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rpt_timed_frames": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "rpt_scatter_colour_plane": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rpt_scatter_colour_plane_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rpt_set_tile_pattern": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rpt_scatter_helper_planes3_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t]),
    "rpt_pack_colour_plane3_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rpt_scatter_colour_plane3_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t]),
    "rpt_colour_plane_ptr": (C.c_void_p, [C.c_void_p]),
    "rpt_set_plane_output": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpt_timing_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "rpt_timing_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "rpt_timing_end_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "rpt_read_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rpt_read_wave_times": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rpt_build_octree": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rpt_free_host": (None, [C.c_void_p]),
    "rpt_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "rpt_version": (C.c_char_p, []),
}


def hip_lib_path() -> str:
    """librpt_hip.so; RPT_HIP_LIB names another build of it (the diagnostics build librpt_hip_diag.so, for the tools)."""
    return os.environ.get("RPT_HIP_LIB") or _path("librpt_hip.so")


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Whichever copy is loaded
    first serves the whole process; if ours pulls in the system copy first, a later ``import torch`` finds no
    GPU.  So when torch is installed, load ITS runtime first (without importing torch) and let librpt_hip.so bind to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


_hip_diag_lib = None


def hip_diag() -> C.CDLL:
    """Load librpt_hip_diag.so — the diagnostics build (`make -C relativitypathtracer_amd/csrc diag`): the product kernels plus
    the instrumented ones and the measurement arms.  For tools/ and tests/test_gpu_diag_arms.py only; raises if it is missing."""
    global _hip_diag_lib
    if _hip_diag_lib is None:
        p = _path("librpt_hip_diag.so")
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: build it with `make -C relativitypathtracer_amd/csrc diag`")
        _share_torch_hip_runtime()
        lib = C.CDLL(p)
        for name, (res, args) in HIP_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _hip_diag_lib = lib
    return _hip_diag_lib


def hip() -> C.CDLL:
    """Load librpt_hip.so (the product render path).  Raises if it is missing — there is no fallback."""
    global _hip_lib
    if _hip_lib is None:
        p = hip_lib_path()
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: the HIP render path has no CPU fallback; build it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'`")
        _share_torch_hip_runtime()
        lib = C.CDLL(p)
        for name, (res, args) in HIP_SYMBOLS.items():
            fn = getattr(lib, name)   # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _hip_lib = lib
    return _hip_lib
