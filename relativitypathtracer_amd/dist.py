"""Multi-GPU frame sharding: pixel-row tiles across ranks + one gather.

The reference is single-device.  Pixels are independent, scene buffers are small and read-only, so
the path shards by replication: every rank holds the whole scene, renders the interleaved 8-row
tiles ``rank, rank+N, rank+2N, ...`` (interleaving balances the expensive mesh rows) into a compact
4 B/pixel colour plane — the x,y floats of the 16 B pixel are constants of the resolution, only the
packed colour changes — and ONE gather (RCCL over xGMI, ``torch.distributed`` backend "nccl") brings
the planes to rank 0, where ``rpt_scatter_colour_plane`` expands them into the reference's 16 B/pixel
framebuffer.  No other collective is on the data path.

The tile arithmetic below is pure Python/numpy so that it can be exercised on CPU (gloo) without a GPU.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

TILE_ROWS = 8


def tile_count(height: int) -> int:
    return (height + TILE_ROWS - 1) // TILE_ROWS


def local_tile_count(height: int, rank: int, world: int) -> int:
    tiles = tile_count(height)
    return 0 if rank >= tiles else (tiles - rank + world - 1) // world


def max_local_tiles(height: int, world: int) -> int:
    return (tile_count(height) + world - 1) // world


def plane_words(width: int, height: int, world: int) -> int:
    """Words (u32) of one rank's padded colour plane: every rank sends the same count."""
    return max_local_tiles(height, world) * TILE_ROWS * width


def pack_plane3(plane_words: np.ndarray) -> np.ndarray:
    """Host restatement of rpt_pack_plane3_kernel: packed R,G,B,1 words -> 3 bytes per pixel."""
    return np.ascontiguousarray(np.asarray(plane_words, dtype=np.uint32).view(np.uint8).reshape(-1, 4)[:, :3]).reshape(-1)


def unpack_plane3(plane_bytes: np.ndarray) -> np.ndarray:
    """3 bytes per pixel -> packed words with the constant alpha byte 1 (what rpt_scatter_plane3_kernel rebuilds)."""
    rgb = np.asarray(plane_bytes, dtype=np.uint8).reshape(-1, 3)
    out = np.empty((rgb.shape[0], 4), dtype=np.uint8)
    out[:, :3] = rgb
    out[:, 3] = 1
    return out.view(np.uint32).reshape(-1)


def tile_rows_of_rank(height: int, rank: int, world: int) -> List[range]:
    """Global row ranges (clipped to the image) rendered by `rank`, in local-tile order."""
    out = []
    for k in range(local_tile_count(height, rank, world)):
        t = rank + k * world
        out.append(range(t * TILE_ROWS, min((t + 1) * TILE_ROWS, height)))
    return out


def extract_plane(packed_frame: np.ndarray, width: int, height: int, rank: int, world: int) -> np.ndarray:
    """The padded plane rank `rank` would send, cut out of a full frame of packed colours [H, W]."""
    plane = np.zeros((max_local_tiles(height, world) * TILE_ROWS, width), dtype=np.uint32)
    for k, rows in enumerate(tile_rows_of_rank(height, rank, world)):
        plane[k * TILE_ROWS:k * TILE_ROWS + len(rows)] = packed_frame[rows.start:rows.stop]
    return plane.reshape(-1)


def reassemble_planes(planes: np.ndarray, width: int, height: int, world: int) -> np.ndarray:
    """Host restatement of rpt_scatter_plane_kernel: [world, plane_words] -> 16 B/pixel framebuffer."""
    from .renderer import PIXEL_DTYPE
    planes = np.asarray(planes, dtype=np.uint32).reshape(world, -1)
    out = np.zeros(height * width, dtype=PIXEL_DTYPE)
    xs = np.arange(width, dtype=np.float32)
    for y in range(height):
        tile = y // TILE_ROWS
        rank, local_row = tile % world, (tile // world) * TILE_ROWS + (y % TILE_ROWS)
        row = planes[rank, local_row * width:(local_row + 1) * width]
        sl = slice(y * width, (y + 1) * width)
        out["x"][sl] = xs
        out["y"][sl] = np.float32(y)
        out["rgba"][sl] = row.view(np.uint8).reshape(width, 4)
    return out


class _Slot:
    """One frame in flight: a context (rpt_ctx) on its own stream with its own output buffers."""
    __slots__ = ("r", "stream", "framebuffer", "plane", "plane3", "gathered", "work", "scattered", "frames")


class FrameSharder:
    """Owns the output tensors of one rank and runs render (+ gather + scatter) frame after frame.

    Frames in flight.  A frame's critical path is the serial octree walk of its dearest pixel, so
    one frame alone leaves most of the GPU idle for most of its duration.  The sharder therefore keeps
    ``len(renderers)`` frames in flight: every slot is a context of its own (scene resident once per context) on
    its own stream with its own output buffers, frame f runs in slot ``f mod depth``, and kernels of consecutive
    frames overlap on the device.  Each frame still gets its own ``rpt_set_objects`` (the host may change
    ``Object[]`` between any two frames) and is rendered completely; ``framebuffer`` is the last submitted frame.

    With more than one rank the three stages of a frame run on different queues — render on the slot's
    stream, the gather on RCCL's stream, the root's reassembly on a side stream — with per-slot planes.
    Per frame every rank still issues exactly one collective, in the same order on all ranks; the host
    never blocks (buffer reuse is ordered by stream-level waits only).
    """

    def __init__(self, renderers, width: int, height: int, rank: int, world: int, force_gather: bool = False,
                 pipeline: bool = True, device=None, plane_bytes: int = 3):
        """`device`: where the output tensors live; default the current GPU.  A CPU device (tests/test_dist_gloo.py:
        gloo, stand-in renderers) runs the same slot rotation and exchange without streams.
        `plane_bytes`: bytes per pixel on the wire — 3 (default: the alpha byte of a packed colour is the constant
        1, so a small kernel drops it before the gather) or 4 (the rendered plane as it is)."""
        import torch
        if not isinstance(renderers, (list, tuple)):
            renderers = [renderers]
        if not pipeline:
            renderers = renderers[:1]
        self.W, self.H, self.rank, self.world = width, height, rank, world
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.on_gpu = dev.type == "cuda"
        self.local_rows = local_tile_count(height, rank, world) * TILE_ROWS
        self.exchange = world > 1 or force_gather      # force_gather: run the plane/gather/scatter path with one rank
        self.depth = len(renderers)
        assert plane_bytes in (3, 4)
        self.plane_bytes = plane_bytes
        self.pipeline = pipeline
        self.frame = 0
        self.last = None
        self.slots = []
        words = plane_words(width, height, world)
        for r in renderers:
            s = _Slot()
            s.r = r
            # The render kernels, the tensors below and what RCCL synchronises with must share ONE real stream per
            # slot.  torch's default stream has handle 0, which the C-ABI reads as "the context's own stream": never it.
            s.stream = torch.cuda.Stream(device=dev) if self.on_gpu else None
            if self.on_gpu:
                r.set_stream(s.stream.cuda_stream)
            s.framebuffer = s.plane = s.plane3 = s.gathered = s.work = s.scattered = None
            s.frames = 0                         # frames submitted to this slot
            if not self.exchange:
                r.set_rows(0, 1, False)
                s.framebuffer = torch.zeros(width * height * 4, dtype=torch.int32, device=dev)
                r.set_output(s.framebuffer.data_ptr())
            else:
                r.set_rows(rank, world, True)
                s.plane = torch.zeros(words, dtype=torch.int32, device=dev)
                r.set_plane_output(s.plane.data_ptr())
                if plane_bytes == 3:
                    s.plane3 = torch.zeros(words * 3, dtype=torch.uint8, device=dev)
                if rank == 0:
                    s.gathered = (torch.zeros((world, words), dtype=torch.int32, device=dev) if plane_bytes == 4 else
                                  torch.zeros((world, words * 3), dtype=torch.uint8, device=dev))
            self.slots.append(s)
        self.r = self.slots[0].r
        self._root_fb = None
        self.side = None
        if self.exchange and rank == 0:
            self._root_fb = torch.zeros(width * height * 4, dtype=torch.int32, device=dev)
            self.side = torch.cuda.Stream(device=dev) if (pipeline and self.on_gpu) else None
        if self.on_gpu:
            torch.cuda.synchronize(dev)     # the zero fills above ran on torch's stream; the slots launch on their own

    @property
    def framebuffer(self):
        """Device tensor holding the most recently submitted frame (16 B/pixel); complete after a device sync."""
        if self.exchange:
            return self._root_fb
        return (self.last or self.slots[0]).framebuffer

    def render_and_gather(self, objects=None):
        """Submit one frame: Object[] refresh (if given: a Scene or raw bytes), render, and with N > 1 the exchange."""
        import torch
        slot = self.slots[self.frame % self.depth]
        self.frame += 1
        slot.frames += 1
        self.last = slot
        if not self.exchange:                 # the context launches on the slot's stream itself: no torch state to switch
            if objects is not None:
                slot.r.set_objects(objects)
            slot.r.render_async()
            return
        if not self.on_gpu:
            if objects is not None:
                slot.r.set_objects(objects)
            self._render_and_gather(slot)
            return
        with torch.cuda.stream(slot.stream):
            if objects is not None:
                slot.r.set_objects(objects)
            self._render_and_gather(slot)

    def _render_and_gather(self, slot):
        import torch
        import torch.distributed as td
        if slot.work is not None:
            slot.work.wait()                        # stream-level on the GPU: this slot's plane has left the device
        if self.rank == 0 and slot.scattered is not None:
            torch.cuda.current_stream().wait_event(slot.scattered)   # this slot's gather buffer has been consumed by the reassembly
        slot.r.render_async()
        send = slot.plane
        if self.plane_bytes == 3:                   # drop the constant alpha byte: 3/4 of the bytes on the wire
            slot.r.pack_colour_plane3(slot.plane.data_ptr(), slot.plane3.data_ptr(), slot.plane.numel(),
                                      stream=slot.stream.cuda_stream if self.on_gpu else None)
            send = slot.plane3
        glist = list(slot.gathered.unbind(0)) if self.rank == 0 else None
        work = td.gather(send, glist, dst=0, async_op=True)      # the one exchange step of the frame
        slot.work = work
        scatter = slot.r.scatter_colour_plane if self.plane_bytes == 4 else slot.r.scatter_colour_plane3
        if self.rank == 0:
            if self.side is None:
                work.wait()
                scatter(slot.gathered.data_ptr(), self._root_fb.data_ptr(), self.W, self.H, self.world, slot.gathered.shape[1])
            else:
                with torch.cuda.stream(self.side):
                    work.wait()
                    scatter(slot.gathered.data_ptr(), self._root_fb.data_ptr(), self.W, self.H, self.world,
                            slot.gathered.shape[1], stream=self.side.cuda_stream)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    slot.scattered = ev
        elif not self.pipeline:
            work.wait()
