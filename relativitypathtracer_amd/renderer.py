"""Renderer — Python face of the C-ABI render path (include/rpt.h, librpt_hip.so).

Mirrors the reference's device-runtime glue (CLSetup.h:22-26): ``initOpenCL`` -> ``Renderer()``,
the eight buffer uploads of main.cpp:33-55 -> ``upload_scene``, ``initCLKernel`` ->
``set_params``/``set_output``, the per-frame ``enqueueWriteBuffer(cl_objects)`` -> ``set_objects``,
``runKernel`` -> ``render``.  Everything executes in the HIP library; there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from .scene import Scene

PIXEL_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("rgba", "u1", (4,)), ("unspecified", "<u4")])
TILE_ROWS = 8


class RenderError(RuntimeError):
    pass


class Renderer:
    def __init__(self, device: int = 0, diag: bool = False):
        self._lib = _ffi.hip_diag() if diag else _ffi.hip()      # diag: librpt_hip_diag.so (measurement arms; tools and tests only)
        h = C.c_void_p()
        rc = self._lib.rpt_create(C.byref(h), int(device))
        if rc != 0 or not h:
            raise RenderError(f"rpt_create(device={device}) failed with status {rc}: no usable gfx950 device "
                              "(the render path has no CPU fallback)")
        self._h = h
        self.device = device
        self.width = self.height = 0
        self._rows = (0, 1, False)
        self._run = 1

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.rpt_destroy(h)

    __del__ = close

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RenderError(f"{what} failed ({rc}): {self._lib.rpt_last_error(self._h).decode()}")

    # -- reference enqueue sequence ------------------------------------------------------------
    def upload_scene(self, scene: Scene):
        d = scene.desc()
        self._check(self._lib.rpt_upload_scene(self._h, C.byref(d)), "rpt_upload_scene")

    def share_scene(self, owner: "Renderer"):
        """Use the scene resident in `owner` (same device) instead of uploading a second copy: frames in flight."""
        self._check(self._lib.rpt_share_scene(self._h, owner._h), "rpt_share_scene")

    def upload_desc(self, desc: _ffi.SceneDesc):
        self._check(self._lib.rpt_upload_scene(self._h, C.byref(desc)), "rpt_upload_scene")

    def set_objects(self, scene_or_bytes):
        if isinstance(scene_or_bytes, Scene):
            d = scene_or_bytes.desc()
            self._check(self._lib.rpt_set_objects(self._h, d.objects, d.object_count), "rpt_set_objects")
        else:
            raw = np.ascontiguousarray(scene_or_bytes).view(np.uint8)
            assert raw.size % 320 == 0
            self._check(self._lib.rpt_set_objects(self._h, raw.ctypes.data, raw.size // 320), "rpt_set_objects")

    def set_params(self, white_point: Sequence[float], ambient: float, width: int, height: int, interval: int):
        wp = (C.c_float * 3)(*white_point)
        self._check(self._lib.rpt_set_params(self._h, wp, float(ambient), int(width), int(height), int(interval)),
                    "rpt_set_params")
        self.width, self.height = int(width), int(height)

    def set_scene_params(self, scene: Scene, width: int, height: int):
        p = scene.params
        self.set_params(p["white_point"], p["ambient"], width, height, p["interval"])

    def set_output(self, device_ptr: Optional[int]):
        self._check(self._lib.rpt_set_output(self._h, C.c_void_p(device_ptr or 0)), "rpt_set_output")

    def set_rows(self, first_tile: int = 0, tile_step: int = 1, colour_plane: bool = False):
        self._check(self._lib.rpt_set_rows(self._h, first_tile, tile_step, int(colour_plane)), "rpt_set_rows")
        self._rows = (first_tile, tile_step, colour_plane)
        self._run = 1

    def set_tile_pattern(self, first_tile: int, tile_step: int, run: int, colour_plane: bool = False):
        """Per period of `tile_step` tiles, the `run` (power of two) consecutive tiles from first_tile on (rpt_set_rows: run 1)."""
        self._check(self._lib.rpt_set_tile_pattern(self._h, first_tile, tile_step, run, int(colour_plane)), "rpt_set_tile_pattern")
        self._rows = (first_tile, tile_step, colour_plane)
        self._run = run

    def set_plane_output(self, device_ptr: Optional[int]):
        self._check(self._lib.rpt_set_plane_output(self._h, C.c_void_p(device_ptr or 0)), "rpt_set_plane_output")

    def set_stream(self, hip_stream: Optional[int]):
        self._check(self._lib.rpt_set_stream(self._h, C.c_void_p(hip_stream or 0)), "rpt_set_stream")

    def set_debug_rgb(self, enable: bool = True):
        self._check(self._lib.rpt_set_debug_rgb(self._h, C.c_void_p(1 if enable else 0)), "rpt_set_debug_rgb")

    def set_variant(self, variant: int):
        self._check(self._lib.rpt_set_variant(self._h, int(variant)), "rpt_set_variant")

    def set_msaa(self, samples_per_axis: int):
        """MSAASAMPLES of opencl_kernel.cl:7 (1 = the reference as shipped)."""
        self._check(self._lib.rpt_set_msaa(self._h, int(samples_per_axis)), "rpt_set_msaa")

    def last_variant(self) -> int:
        """The kernel variant (include/rpt.h) the last launch of this context was made with: what variant 0 resolved to."""
        return int(self._lib.rpt_last_variant(self._h))

    def verify_frame(self) -> int:
        """Pixels whose packed colour differs between the kernel a frame would get and the un-culled kernel (0 = the culls changed nothing)."""
        n = C.c_uint64(0)
        self._check(self._lib.rpt_verify_frame(self._h, C.byref(n)), "rpt_verify_frame")
        return int(n.value)

    def render(self):
        self._check(self._lib.rpt_render(self._h), "rpt_render")

    def render_async(self):
        self._check(self._lib.rpt_render_async(self._h), "rpt_render_async")

    def sync(self):
        self._check(self._lib.rpt_sync(self._h), "rpt_sync")

    # -- results -------------------------------------------------------------------------------
    def local_tiles(self) -> int:
        first, step, _ = self._rows
        tiles = (self.height + TILE_ROWS - 1) // TILE_ROWS
        if first >= tiles:
            return 0
        full, rest = divmod(tiles - first, step)
        return full * self._run + min(rest, self._run)

    def output_ptr(self) -> int:
        return self._lib.rpt_output_ptr(self._h) or 0

    def colour_plane_ptr(self) -> int:
        return self._lib.rpt_colour_plane_ptr(self._h) or 0

    def read_framebuffer(self) -> np.ndarray:
        """16 B/pixel framebuffer as a structured array [height*width] (row 0 = bottom row)."""
        out = np.empty(self.width * self.height, dtype=PIXEL_DTYPE)
        self._check(self._lib.rpt_read_framebuffer(self._h, out.ctypes.data, out.nbytes), "rpt_read_framebuffer")
        return out

    def read_colour_plane(self) -> np.ndarray:
        out = np.empty((self.local_tiles() * TILE_ROWS, self.width), dtype=np.uint32)
        self._check(self._lib.rpt_read_framebuffer(self._h, out.ctypes.data, out.nbytes), "rpt_read_framebuffer")
        return out

    def read_debug_rgb(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._check(self._lib.rpt_read_debug_rgb(self._h, out.ctypes.data, out.nbytes), "rpt_read_debug_rgb")
        return out

    def last_frame_ms(self) -> float:
        ms = C.c_float()
        self._check(self._lib.rpt_last_frame_ms(self._h, C.byref(ms)), "rpt_last_frame_ms")
        return ms.value

    def timed_frames(self, frames: int) -> float:
        ms = C.c_float()
        self._check(self._lib.rpt_timed_frames(self._h, int(frames), C.byref(ms)), "rpt_timed_frames")
        return ms.value

    def scatter_colour_plane(self, planes_ptr: int, out16_ptr: int, width: int, height: int, n_ranks: int,
                             plane_stride_words: int, stream: Optional[int] = None):
        self._check(self._lib.rpt_scatter_colour_plane_on(self._h, C.c_void_p(stream or 0), C.c_void_p(planes_ptr),
                                                          C.c_void_p(out16_ptr), width, height, n_ranks, plane_stride_words),
                    "rpt_scatter_colour_plane")

    def pack_colour_plane3(self, plane4_ptr: int, plane3_ptr: int, pixels: int, stream: Optional[int] = None):
        """4 B/pixel colour plane -> 3 B/pixel (the alpha byte is the constant 1), on `stream` or the launch stream."""
        self._check(self._lib.rpt_pack_colour_plane3_on(self._h, C.c_void_p(stream or 0), C.c_void_p(plane4_ptr), C.c_void_p(plane3_ptr),
                                                        int(pixels)), "rpt_pack_colour_plane3_on")

    def scatter_helper_planes3(self, planes3_ptr: int, out16_ptr: int, width: int, height: int, n_ranks: int, root_run: int,
                               stride_bytes: int, stream: Optional[int] = None):
        self._check(self._lib.rpt_scatter_helper_planes3_on(self._h, C.c_void_p(stream or 0), C.c_void_p(planes3_ptr), C.c_void_p(out16_ptr),
                                                            width, height, n_ranks, root_run, int(stride_bytes)), "rpt_scatter_helper_planes3_on")

    def scatter_colour_plane3(self, planes3_ptr: int, out16_ptr: int, width: int, height: int, n_ranks: int, stride_bytes: int,
                              stream: Optional[int] = None):
        self._check(self._lib.rpt_scatter_colour_plane3_on(self._h, C.c_void_p(stream or 0), C.c_void_p(planes3_ptr), C.c_void_p(out16_ptr),
                                                           width, height, n_ranks, int(stride_bytes)), "rpt_scatter_colour_plane3_on")

    def read_counters(self):
        out = (C.c_uint64 * 16)()
        self._check(self._lib.rpt_read_counters(self._h, out), "rpt_read_counters")
        return list(out)

    def read_wave_times(self) -> np.ndarray:
        n = ((self.width + 31) // 32) * self.local_tiles() * 4 * 10
        out = np.zeros(n, dtype=np.uint64)
        got = C.c_size_t()
        self._check(self._lib.rpt_read_wave_times(self._h, out.ctypes.data, n, C.byref(got)), "rpt_read_wave_times")
        return out[:got.value].reshape(-1, 10)

    def build_octree(self, scene: Scene, first_triangle_word: int):
        """GPU octree build for the geometry `scene.ReadOBJ(..., octree=False)` just imported; appends it to the scene."""
        d = scene.desc()
        nodes, tris = C.c_void_p(), C.c_void_p()
        n_nodes, n_tris = C.c_size_t(), C.c_size_t()
        self._check(self._lib.rpt_build_octree(self._h, d.vertices, d.vertex_count, d.triangles, d.triangle_words,
                                               first_triangle_word, d.octree_count, d.octree_tri_count,
                                               C.byref(nodes), C.byref(n_nodes), C.byref(tris), C.byref(n_tris)),
                    "rpt_build_octree")
        try:
            scene.append_octree(nodes.value, n_nodes.value, tris.value, n_tris.value)
        finally:
            self._lib.rpt_free_host(nodes)
            self._lib.rpt_free_host(tris)

    def probe(self, which: int, inputs: np.ndarray, out_width: int) -> np.ndarray:
        inputs = np.ascontiguousarray(inputs, dtype=np.float32)
        n = inputs.shape[0]
        out = np.empty((n, out_width), dtype=np.float32)
        self._check(self._lib.rpt_probe(self._h, which, inputs.ctypes.data, out.ctypes.data, n), "rpt_probe")
        return out

    def probe_division(self, mode: int, seed: int, blocks: int, per_thread: int, max_samples: int = 16):
        """rpt_probe_division (include/rpt.h): (counts[5], samples[max_samples, 4])."""
        counts = (C.c_uint64 * 5)()
        samples = np.zeros((max_samples, 4), dtype=np.float32)
        self._check(self._lib.rpt_probe_division(self._h, int(mode), int(seed) & 0xffffffff, int(blocks), int(per_thread), counts, samples.ctypes.data, max_samples), "rpt_probe_division")
        return [int(c) for c in counts], samples

    def probe_object(self, which: int, object_index: int, inputs: np.ndarray) -> np.ndarray:
        """rpt_probe_object (include/rpt.h): which = 0 (n, 8) rest-frame rays -> (n, 8); 1 (n, 9) shadow rays -> (n, 2) {un-culled, culled}
        occlusion; 2 (n, 4) vectors -> (n, 16) the four transforms; 3 (n, 3) camera directions -> (n, 8)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.float32)
        n = inputs.shape[0]
        out = np.empty((n, {0: 8, 1: 2, 2: 16, 3: 8}[which]), dtype=np.float32)
        self._check(self._lib.rpt_probe_object(self._h, int(which), int(object_index), inputs.ctypes.data, out.ctypes.data, n), "rpt_probe_object")
        return out

    def mesh_segment_cull_record(self, object_index: int) -> np.ndarray:
        out = np.zeros(10, dtype=np.float32)
        self._check(self._lib.rpt_mesh_segment_cull_record(self._h, int(object_index), out.ctypes.data_as(C.POINTER(C.c_float))), "rpt_mesh_segment_cull_record")
        return out

    def probe_walk(self, object_index: int, rays: np.ndarray) -> np.ndarray:
        """rays (n, 6) = object-space origin and direction -> (n, 3, 8): {hit, dist, normal.xyz, uv.xy, 0} from the reference-layout
        walk, the throughput walk and the latency walk (include/rpt.h, rpt_probe_walk)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = rays.shape[0]
        out = np.empty((n, 3, 8), dtype=np.float32)
        self._check(self._lib.rpt_probe_walk(self._h, int(object_index), rays.ctypes.data, out.ctypes.data, n), "rpt_probe_walk")
        return out


def render_scene(scene: Scene, width: int, height: int, device: int = 0, debug_rgb: bool = False):
    """Convenience: upload, render one frame, read back. Returns (pixels, rgb-or-None)."""
    r = Renderer(device)
    try:
        r.upload_scene(scene)
        r.set_scene_params(scene, width, height)
        r.set_output(None)
        if debug_rgb:
            r.set_debug_rgb(True)
        r.render()
        return r.read_framebuffer(), (r.read_debug_rgb() if debug_rgb else None)
    finally:
        r.close()
