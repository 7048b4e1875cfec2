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


class FrameSharder:
    """Owns the output tensors of one rank and runs render (+ gather + scatter) frame after frame.

    With more than one rank the three stages of a frame run on different queues — render on the launch
    stream, the gather on RCCL's stream, the root's reassembly on a side stream — and consecutive frames
    overlap two deep (double-buffered planes): frame k+1 renders while frame k's plane is on the wire.
    Per frame every rank still issues exactly one collective, in the same order on all ranks; the host
    never blocks (buffer reuse is ordered by stream-level waits only).
    """

    def __init__(self, renderer, width: int, height: int, rank: int, world: int, force_gather: bool = False,
                 pipeline: bool = True):
        import torch
        self.r, self.W, self.H, self.rank, self.world = renderer, width, height, rank, world
        dev = torch.device("cuda", torch.cuda.current_device())
        self.framebuffer: Optional["torch.Tensor"] = None
        self.local_rows = local_tile_count(height, rank, world) * TILE_ROWS
        self.exchange = world > 1 or force_gather      # force_gather: run the plane/gather/scatter path with one rank
        self.depth = 2 if pipeline else 1
        self.frame = 0
        # The render kernels, the tensors below and what RCCL synchronises with must share ONE real stream.  torch's
        # default stream has handle 0, which the C-ABI reads as "the context's own stream": never use it here.
        self.stream = torch.cuda.current_stream()
        if self.stream.cuda_stream == 0:
            self.stream = torch.cuda.Stream(device=dev)
        renderer.set_stream(self.stream.cuda_stream)
        if not self.exchange:
            renderer.set_rows(0, 1, False)
            self.framebuffer = torch.zeros(width * height * 4, dtype=torch.int32, device=dev)
            renderer.set_output(self.framebuffer.data_ptr())
            return
        words = plane_words(width, height, world)
        renderer.set_rows(rank, world, True)
        self.planes = [torch.zeros(words, dtype=torch.int32, device=dev) for _ in range(self.depth)]
        self.works = [None] * self.depth            # gather of the frame that last used plane slot i
        if rank == 0:
            self.gathered = [torch.zeros((world, words), dtype=torch.int32, device=dev) for _ in range(self.depth)]
            self.framebuffer = torch.zeros(width * height * 4, dtype=torch.int32, device=dev)
            self.side = torch.cuda.Stream(device=dev) if pipeline else None
            self.scattered = [None] * self.depth    # event: reassembly of the frame that last used gathered slot i

    def render_and_gather(self):
        import torch
        with torch.cuda.stream(self.stream):
            self._render_and_gather()

    def _render_and_gather(self):
        if not self.exchange:
            self.r.render_async()
            return
        import torch
        import torch.distributed as td
        slot = self.frame % self.depth
        self.frame += 1
        cur = torch.cuda.current_stream()
        if self.works[slot] is not None:
            self.works[slot].wait()                 # stream-level: this plane slot has left the GPU
        if self.rank == 0 and self.scattered[slot] is not None:
            cur.wait_event(self.scattered[slot])    # this gather slot has been consumed by the reassembly
        self.r.set_plane_output(self.planes[slot].data_ptr())
        self.r.render_async()
        glist = list(self.gathered[slot].unbind(0)) if self.rank == 0 else None
        work = td.gather(self.planes[slot], glist, dst=0, async_op=True)      # the one exchange step of the frame
        self.works[slot] = work
        if self.rank == 0:
            if self.side is None:
                work.wait()
                self.r.scatter_colour_plane(self.gathered[slot].data_ptr(), self.framebuffer.data_ptr(), self.W, self.H,
                                            self.world, self.gathered[slot].shape[1])
            else:
                with torch.cuda.stream(self.side):
                    work.wait()
                    self.r.scatter_colour_plane(self.gathered[slot].data_ptr(), self.framebuffer.data_ptr(), self.W, self.H,
                                                self.world, self.gathered[slot].shape[1], stream=self.side.cuda_stream)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.scattered[slot] = ev
        elif self.depth == 1:
            work.wait()
