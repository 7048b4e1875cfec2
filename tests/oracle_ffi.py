"""ctypes binding of oracle/librpt_oracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librpt_oracle.so")


class OracleArgs(C.Structure):
    _fields_ = [
        ("objects", C.c_void_p), ("object_count", C.c_int32),
        ("vertices", C.c_void_p), ("normals", C.c_void_p), ("uvs", C.c_void_p),
        ("triangles", C.c_void_p), ("octrees", C.c_void_p), ("octreeTris", C.c_void_p),
        ("textures", C.c_void_p), ("texture_bytes", C.c_uint64),
        ("white_point", C.c_float * 3), ("ambient", C.c_float),
        ("width", C.c_int32), ("height", C.c_int32), ("interval", C.c_int32),
        ("out_pixels", C.c_void_p), ("out_rgb", C.c_void_p), ("msaa", C.c_int32),
    ]


STAT_FIELDS = ["shadow_rays", "sphere_tests", "cube_tests", "octree_calls", "root_aabb_hits", "inside_starts",
               "inside_descent_steps", "descent_steps", "leaf_visits", "tri_tests", "pixels_hit",
               "distinct_tri_tests", "repeats_of_previous_leaf"]


class OracleStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS]


_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        l = C.CDLL(ORACLE_SO)
        l.rpt_oracle_render.restype = C.c_int
        l.rpt_oracle_render.argtypes = [C.POINTER(OracleArgs), C.c_int, C.c_int, C.c_int, C.POINTER(OracleStats)]
        FP = C.POINTER(C.c_float)
        l.rpt_oracle_tri.restype = C.c_int
        l.rpt_oracle_tri.argtypes = [FP, FP, FP, FP, FP, FP]
        l.rpt_oracle_aabb.restype = C.c_int
        l.rpt_oracle_aabb.argtypes = [FP, FP, FP, FP, FP, C.POINTER(C.c_int)]
        l.rpt_oracle_camray.restype = None
        l.rpt_oracle_camray.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, FP]
        l.rpt_oracle_hable.restype = None
        l.rpt_oracle_hable.argtypes = [FP, FP]
        l.rpt_oracle_walk_steps.restype = None
        l.rpt_oracle_walk_steps.argtypes = [FP, FP, FP, FP]
        l.rpt_oracle_octree_rays.restype = C.c_int
        l.rpt_oracle_octree_rays.argtypes = [C.POINTER(OracleArgs), C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        l.rpt_oracle_object_rays.restype = C.c_int
        l.rpt_oracle_object_rays.argtypes = [C.POINTER(OracleArgs), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        l.rpt_oracle_asin_atan2.restype = None
        l.rpt_oracle_asin_atan2.argtypes = [C.c_float, C.c_float, C.c_float, FP]
        _lib = l
    return _lib


def render(scene, width: int, height: int, *, rows=None, threads: int = 0, want_rgb: bool = True,
           want_stats: bool = False, objects: np.ndarray | None = None, interval: int | None = None, msaa: int = 1):
    """Render `scene` (a relativitypathtracer_amd.Scene whose objects are up to date) on the CPU oracle.

    Returns (pixels[H*W] structured 16 B, rgb[H,W,3] float32 or None, stats dict or None).
    Rows outside `rows` are left zero.
    """
    d = scene.desc()
    p = scene.params
    a = OracleArgs()
    if objects is not None:
        objects = np.ascontiguousarray(objects).view(np.uint8)
        a.objects, a.object_count = objects.ctypes.data, objects.size // 320
    else:
        a.objects, a.object_count = d.objects, d.object_count
    a.vertices, a.normals, a.uvs = d.vertices, d.normals, d.uvs
    a.triangles, a.octrees, a.octreeTris = d.triangles, d.octrees, d.octreeTris
    a.textures, a.texture_bytes = d.textures, d.texture_bytes
    a.white_point = (C.c_float * 3)(*p["white_point"])
    a.ambient = p["ambient"]
    a.width, a.height = width, height
    a.interval = p["interval"] if interval is None else interval
    a.msaa = msaa
    pixels = np.zeros(width * height, dtype=PIXEL_DTYPE)
    a.out_pixels = pixels.ctypes.data
    rgb = None
    if want_rgb:
        rgb = np.zeros((height, width, 3), dtype=np.float32)
        a.out_rgb = rgb.ctypes.data
    st = OracleStats() if want_stats else None
    r0, r1 = (0, height) if rows is None else rows
    if threads <= 0:
        threads = os.cpu_count() or 1
    rc = lib().rpt_oracle_render(C.byref(a), r0, r1, threads, C.byref(st) if st is not None else None)
    if rc != 0:
        raise RuntimeError(f"rpt_oracle_render failed: {rc}")
    stats = {n: getattr(st, n) for n in STAT_FIELDS} if st is not None else None
    return pixels, rgb, stats


def octree_rays(scene, object_index: int, rays: np.ndarray) -> np.ndarray:
    """opencl_kernel.cl:206-306 on object-space rays (n, 6) through mesh object `object_index`: (n, 8) = hit, dist, normal.xyz, uv.xy, 0."""
    d = scene.desc()
    a = OracleArgs()
    a.objects, a.object_count = d.objects, d.object_count
    a.vertices, a.normals, a.uvs = d.vertices, d.normals, d.uvs
    a.triangles, a.octrees, a.octreeTris = d.triangles, d.octrees, d.octreeTris
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    out = np.empty((rays.shape[0], 8), dtype=np.float32)
    rc = lib().rpt_oracle_octree_rays(C.byref(a), int(object_index), rays.ctypes.data, out.ctypes.data, rays.shape[0])
    if rc != 0:
        raise RuntimeError(f"rpt_oracle_octree_rays failed: {rc}")
    return out


def object_rays(scene, which: int, object_index: int, inputs: np.ndarray, objects: np.ndarray | None = None) -> np.ndarray:
    """rpt_oracle_object_rays: which = 0 (n, 8) rest-frame rays through one object's intersector -> (n, 8); 1 (n, 9) shadow rays
    {origin4, dir4, lightDist} with light `object_index` -> (n,) first occluder or -1; 2 (n, 4) vectors -> (n, 16) the four transforms."""
    d = scene.desc()
    a = OracleArgs()
    if objects is not None:
        objects = np.ascontiguousarray(objects).view(np.uint8)
        a.objects, a.object_count = objects.ctypes.data, objects.size // 320
    else:
        a.objects, a.object_count = d.objects, d.object_count
    a.vertices, a.normals, a.uvs = d.vertices, d.normals, d.uvs
    a.triangles, a.octrees, a.octreeTris = d.triangles, d.octrees, d.octreeTris
    a.textures, a.texture_bytes = d.textures, d.texture_bytes
    a.interval = scene.params["interval"]
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    n = inputs.shape[0]
    out = np.empty((n, 8) if which == 0 else (n,) if which == 1 else (n, 16), dtype=np.float32)
    rc = lib().rpt_oracle_object_rays(C.byref(a), int(which), int(object_index), inputs.ctypes.data, out.ctypes.data, n)
    if rc != 0:
        raise RuntimeError(f"rpt_oracle_object_rays failed: {rc}")
    return out


PIXEL_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("rgba", "u1", (4,)), ("unspecified", "<u4")])
assert PIXEL_DTYPE.itemsize == 16
