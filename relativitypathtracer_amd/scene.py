"""Host-side scene: a thin Python face on librpt_scene.so.

Mirrors the reference's host interface for the steps either side of the render path —
``inputScene`` / ``ReadOBJ`` / ``ReadTexture`` (Render.cpp:211-538) and the per-frame Lorentz refresh
of ``Object[]`` (Render.cpp:179-200) — with the same names and argument meaning.  All the work is
done in C++ (csrc/host); this module only marshals, and decodes texture files with Pillow (the
reference uses CImg + libjpeg, Render.cpp:418-434; decoders may differ by an LSB, so every consumer
of one Scene sees the same decoded bytes).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import _ffi

ASSET_ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "reference")

# Scenes/bunny.txt names a mesh the reference does not ship (.MISSING_LARGE_BLOBS); BASELINE.md §3
# substitutes Models/bunny.obj.
DEFAULT_ALIASES = {"Models/StanfordBunny.obj": "Models/bunny.obj"}

_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]


def _pil_decoder(path: bytes, rgb_out, w_out, h_out, _user) -> int:
    try:
        from PIL import Image
        with Image.open(path.decode()) as im:
            arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
        h, w = arr.shape[:2]
        buf = _libc.malloc(arr.nbytes)
        if not buf:
            return 1
        C.memmove(buf, arr.ctypes.data, arr.nbytes)
        rgb_out[0] = buf
        w_out[0] = w
        h_out[0] = h
        return 0
    except Exception:
        return 1


_PIL_DECODER = _ffi.TextureDecoder(_pil_decoder)


class SceneError(RuntimeError):
    pass


class Scene:
    """Scene state: objects, velocities, mesh pool, texture pool, camera, scalars."""

    def __init__(self, asset_root: str = ASSET_ROOT, aliases: Optional[Dict[str, str]] = None):
        self._lib = _ffi.scene_lib()
        self._h = self._lib.rpt_scene_create()
        if not self._h:
            raise SceneError("rpt_scene_create failed")
        self._lib.rpt_scene_set_asset_root(self._h, asset_root.encode())
        for k, v in (DEFAULT_ALIASES if aliases is None else aliases).items():
            self._lib.rpt_scene_add_alias(self._h, k.encode(), v.encode())
        self._lib.rpt_scene_set_texture_decoder(self._h, _PIL_DECODER, None)
        self.asset_root = asset_root

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.rpt_scene_destroy(h)

    # -- construction ---------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise SceneError(f"{what}: {self._lib.rpt_scene_last_error(self._h).decode()}")

    def inputScene(self, text: str) -> str:
        """Parse a whole scene description (what the reference reads from stdin). Returns diagnostics."""
        self._check(self._lib.rpt_scene_input(self._h, text.encode()), "inputScene")
        return self._lib.rpt_scene_last_error(self._h).decode()

    def ReadOBJ(self, path: str, octree: bool = True) -> int:
        """Import an OBJ.  octree=False imports the geometry only and returns the word offset of its first triangle,
        for Renderer.build_octree (the GPU builder) to finish."""
        if octree:
            self._check(self._lib.rpt_scene_read_obj(self._h, path.encode()), "ReadOBJ")
            return -1
        first = C.c_size_t()
        self._check(self._lib.rpt_scene_read_obj_geometry(self._h, path.encode(), C.byref(first)), "ReadOBJ")
        return first.value

    def append_octree(self, nodes_ptr: int, node_count: int, tris_ptr: int, tri_count: int):
        self._check(self._lib.rpt_scene_append_octree(self._h, C.c_void_p(nodes_ptr), node_count, C.c_void_p(tris_ptr), tri_count),
                    "append_octree")

    def ReadTexture(self, path: str):
        self._check(self._lib.rpt_scene_read_texture(self._h, path.encode()), "ReadTexture")

    def add_texture(self, rgb: np.ndarray):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        self._check(self._lib.rpt_scene_add_texture_rgb8(self._h, rgb.ctypes.data, rgb.shape[1], rgb.shape[0]),
                    "add_texture")

    @classmethod
    def from_file(cls, name: str, asset_root: str = ASSET_ROOT, aliases: Optional[Dict[str, str]] = None) -> "Scene":
        """Load ``Scenes/<name>.txt`` (or an explicit path) from the asset root."""
        path = name
        if not os.path.exists(path):
            path = os.path.join(asset_root, "Scenes", name if name.endswith(".txt") else name + ".txt")
        with open(path) as f:
            text = f.read()
        s = cls(asset_root, aliases)
        s.inputScene(text)
        return s

    # -- camera / time --------------------------------------------------------------------
    def set_camera(self, velocity: Sequence[float] = (0, 0, 0), t: float = 0.0, position=(0.0, 0.0, 0.0)):
        v = (C.c_float * 3)(*velocity)
        p = (C.c_float * 4)(t, *position)
        self._check(self._lib.rpt_scene_set_camera(self._h, v, p), "set_camera")

    def get_camera(self):
        v = (C.c_float * 3)()
        p = (C.c_float * 4)()
        self._lib.rpt_scene_get_camera(self._h, v, p)
        return list(v), list(p)

    def accelerate(self, direction: Sequence[float], frame_ms: int):
        self._check(self._lib.rpt_scene_accelerate(self._h, (C.c_float * 3)(*direction), int(frame_ms)), "accelerate")

    def reset_velocity(self):
        self._lib.rpt_scene_reset_velocity(self._h)

    def set_paused(self, paused: bool):
        self._lib.rpt_scene_set_paused(self._h, int(paused))

    def advance_time(self, frame_ms: int):
        self._lib.rpt_scene_advance_time(self._h, int(frame_ms))

    def set_interval(self, interval: int):
        self._lib.rpt_scene_set_interval(self._h, int(interval))

    def toggle_interval(self):
        self._lib.rpt_scene_toggle_interval(self._h)

    def update_objects(self):
        """Per-frame refresh of Object.Lorentz / InvLorentz / stationaryCam (Render.cpp:179-200)."""
        self._check(self._lib.rpt_scene_update_objects(self._h), "update_objects")

    # -- views ----------------------------------------------------------------------------
    def desc(self) -> _ffi.SceneDesc:
        d = _ffi.SceneDesc()
        self._check(self._lib.rpt_scene_get_desc(self._h, C.byref(d)), "get_desc")
        return d

    @property
    def params(self):
        wp = (C.c_float * 3)()
        amb = C.c_float()
        itv = C.c_int()
        self._lib.rpt_scene_get_params(self._h, wp, C.byref(amb), C.byref(itv))
        return {"white_point": [wp[0], wp[1], wp[2]], "ambient": amb.value, "interval": itv.value}

    def _view(self, ptr, count, dtype, width=None) -> np.ndarray:
        if not ptr or count == 0:
            shape = (0,) if width is None else (0, width)
            return np.zeros(shape, dtype=dtype)
        n = count if width is None else count * width
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,))
        return arr if width is None else arr.reshape(count, width)

    def buffers(self) -> Dict[str, np.ndarray]:
        """Copies of the eight scene arrays as numpy (bytes exactly as uploaded)."""
        d = self.desc()
        return {
            "objects": self._view(d.objects, d.object_count * 320, np.uint8).copy(),
            "vertices": self._view(d.vertices, d.vertex_count, np.float32, 4).copy(),
            "normals": self._view(d.normals, d.normal_count, np.float32, 4).copy(),
            "uvs": self._view(d.uvs, d.uv_count, np.float32, 2).copy(),
            "triangles": self._view(d.triangles, d.triangle_words, np.uint32).copy(),
            "octrees": self._view(d.octrees, d.octree_count * 96, np.uint8).copy(),
            "octreeTris": self._view(d.octreeTris, d.octree_tri_count, np.int32).copy(),
            "textures": self._view(d.textures, d.texture_bytes, np.uint8).copy(),
        }

    def objects(self) -> np.ndarray:
        """Structured copy of Object[]."""
        d = self.desc()
        raw = self._view(d.objects, d.object_count * 320, np.uint8).copy()
        return raw.view(OBJECT_DTYPE)

    def octrees(self) -> np.ndarray:
        d = self.desc()
        raw = self._view(d.octrees, d.octree_count * 96, np.uint8).copy()
        return raw.view(OCTREE_DTYPE)

    def velocities(self) -> np.ndarray:
        p = C.c_void_p()
        n = C.c_size_t()
        self._lib.rpt_scene_get_velocities(self._h, C.byref(p), C.byref(n))
        return self._view(p.value, n.value, np.float32, 4).copy()

    def mesh_roots(self):
        p = C.c_void_p()
        n = C.c_size_t()
        self._lib.rpt_scene_get_mesh_roots(self._h, C.byref(p), C.byref(n))
        return self._view(p.value, n.value, np.int32).copy().tolist()


OBJECT_DTYPE = np.dtype({
    "names": ["M", "InvM", "Lorentz", "InvLorentz", "stationaryCam", "color", "type", "meshIndex",
              "textureIndex", "textureWidth", "textureHeight", "light", "flashPeriod", "flashDuration"],
    "formats": [("<f4", (4, 4))] * 4 + [("<f4", (4,))] * 2 + ["<i4"] * 5 + ["u1", "<f4", "<f4"],
    "offsets": [0, 64, 128, 192, 256, 272, 288, 292, 296, 300, 304, 308, 312, 316],
    "itemsize": 320,
})

OCTREE_DTYPE = np.dtype({
    "names": ["min", "max", "trisIndex", "trisCount", "children", "neighbors"],
    "formats": [("<f4", (4,)), ("<f4", (4,)), "<i4", "<i4", ("<i4", (8,)), ("<i4", (6,))],
    "offsets": [0, 16, 32, 36, 40, 72],
    "itemsize": 96,
})


def write_ppm(path: str, pixels16: np.ndarray, width: int, height: int):
    """Framebuffer consumer (gl_interop.cpp:51-67 stand-in): 16 B/px framebuffer -> binary PPM, top row first."""
    buf = np.ascontiguousarray(pixels16).view(np.uint8)
    assert buf.size == width * height * 16
    rc = _ffi.scene_lib().rpt_write_ppm(path.encode(), buf.ctypes.data, width, height)
    if rc != 0:
        raise SceneError(f"rpt_write_ppm({path}) failed")


def write_png(path: str, pixels16: np.ndarray, width: int, height: int):
    """The same image as an 8-bit RGB PNG (uncompressed zlib stream, written by the library itself)."""
    buf = np.ascontiguousarray(pixels16).view(np.uint8)
    assert buf.size == width * height * 16
    rc = _ffi.scene_lib().rpt_write_png(path.encode(), buf.ctypes.data, width, height)
    if rc != 0:
        raise SceneError(f"rpt_write_png({path}) failed")
