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


def pattern_tiles(height: int, first: int, step: int, run: int) -> List[int]:
    """Global tiles of a context set with rpt_set_tile_pattern(first, step, run), in local-tile order."""
    out, tiles, t = [], tile_count(height), 0
    while True:
        g = (t // run) * step + first + (t % run)
        if (t // run) * step + first >= tiles:
            break
        if g < tiles:
            out.append(g)
        t += 1
    return out


def weighted_helper_words(width: int, height: int, world: int, root_run: int) -> int:
    """Words (u32) of one helper's padded colour plane in the weighted split: one tile per period of root_run + N - 1."""
    period = root_run + world - 1
    return ((tile_count(height) + period - 1) // period) * TILE_ROWS * width


def choose_root_run(frame_s: float, gather_base_s: float, gather_s_per_byte: float, width: int, height: int, world: int,
                    overhead_s: float = 0.0) -> int:
    """The split that minimises the modelled frame time of a pipelined N-rank job, from three measured numbers:
    `frame_s` (one rank rendering the WHOLE frame, frames in flight), and the exchange's linear cost model
    gather(b bytes per rank) = gather_base_s + gather_s_per_byte * b.  Candidates: the root renders everything and
    nothing is exchanged (returns 0); the weighted split with root_run in 1, 2, 4, 8, 16 (the root renders root_run of
    every root_run + N - 1 tiles straight into the framebuffer, the others one tile each into a 3 B/px plane).
    Stages overlap, so a frame costs the slowest of: the root's render share, a helper's, the exchange."""
    best, best_t = 0, frame_s
    for run in (1, 2, 4, 8, 16):
        period = run + world - 1
        share_root, share_helper = run / period, 1.0 / period
        wire = gather_base_s + gather_s_per_byte * 3 * weighted_helper_words(width, height, world, run)
        t = max(frame_s * share_root + overhead_s, frame_s * share_helper + overhead_s, wire)
        if t < best_t * 0.97:            # a split must clearly beat the simpler arrangement before it
            best, best_t = run, t
    return best


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


def calibrate_split(renderers, objects, width: int, height: int, rank: int, world: int, device=None, frames: int = 30):
    """Measure what choose_root_run needs and agree on the split: every rank renders the whole frame `frames` times
    with len(renderers) frames in flight (rank 0's time counts), all ranks time gathers of the smallest and the
    largest helper plane (a linear cost model of the exchange on THIS node's links), rank 0 picks the split and
    broadcasts it.  Collectives: the timing gathers, one broadcast — the same on every rank.  Returns
    (root_run, info dict)."""
    import time
    import torch
    import torch.distributed as td
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    on_gpu = dev.type == "cuda"

    def sync():
        if on_gpu:
            torch.cuda.synchronize(dev)

    for r in renderers:
        r.set_rows(0, 1, False)
        r.set_output(None)
    for k in range(2 * len(renderers)):
        renderers[k % len(renderers)].set_objects(objects)
        renderers[k % len(renderers)].render_async()
    for r in renderers:
        r.sync()
    t0 = time.perf_counter()
    for k in range(frames):
        renderers[k % len(renderers)].set_objects(objects)
        renderers[k % len(renderers)].render_async()
    for r in renderers:
        r.sync()
    frame_s = (time.perf_counter() - t0) / frames

    def time_gather(nbytes, reps=8):
        send = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        recv = torch.zeros((world, nbytes), dtype=torch.uint8, device=dev) if rank == 0 else None
        for _ in range(2):
            td.gather(send, list(recv.unbind(0)) if rank == 0 else None, dst=0)
        sync()
        td.barrier()
        sync()
        t1 = time.perf_counter()
        works = [td.gather(send, list(recv.unbind(0)) if rank == 0 else None, dst=0, async_op=True) for _ in range(reps)]
        for w in works:
            w.wait()
        sync()
        return (time.perf_counter() - t1) / reps

    b_small = 3 * weighted_helper_words(width, height, world, 16)
    b_big = 3 * weighted_helper_words(width, height, world, 1)
    t_small, t_big = time_gather(b_small), time_gather(b_big)
    per_byte = max((t_big - t_small) / max(b_big - b_small, 1), 0.0)
    base = max(t_small - per_byte * b_small, 0.0)
    choice = torch.zeros(1, dtype=torch.int32, device=dev)
    if rank == 0:
        choice[0] = choose_root_run(frame_s, base, per_byte, width, height, world)
    td.broadcast(choice, src=0)
    sync()
    return int(choice[0]), {"frame_ms_one_rank": round(frame_s * 1e3, 4), "gather_base_ms": round(base * 1e3, 4),
                            "gather_GBps_per_rank": round(1e-9 / per_byte, 2) if per_byte > 0 else None}


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
                 pipeline: bool = True, device=None, plane_bytes: int = 3, root_run: Optional[int] = None):
        """`device`: where the output tensors live; default the current GPU.  A CPU device (tests/test_dist_gloo.py:
        gloo, stand-in renderers) runs the same slot rotation and exchange without streams.
        `plane_bytes`: bytes per pixel on the wire — 3 (default: the alpha byte of a packed colour is the constant
        1, so a small kernel drops it before the gather) or 4 (the rendered plane as it is).
        `root_run`: None = equal interleaved split (tile k -> rank k mod N, every plane gathered).  A power of two =
        the WEIGHTED split: per period of root_run + N - 1 tiles rank 0 renders root_run tiles straight into its
        framebuffer and rank j the single tile root_run + j - 1 into a 3 B/px plane — pixels rendered where they are
        needed cross no link, so the root takes the larger share (choose_root_run sizes it).  0 = rank 0 renders the
        whole frame and nothing is exchanged (the other ranks idle): the arrangement to fall back to when the
        exchange is slower than rendering."""
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
        self.root_run = root_run if ((world > 1 or force_gather) and root_run is not None) else None
        self.solo = self.root_run == 0                  # rank 0 renders everything, no exchange
        self.weighted = bool(self.root_run)
        if self.solo:
            self.exchange = False
        self.depth = len(renderers)
        assert plane_bytes in (3, 4)
        self.plane_bytes = plane_bytes
        self.pipeline = pipeline
        self.frame = 0
        self.last = None
        self.slots = []
        words = plane_words(width, height, world)
        if self.weighted:
            assert plane_bytes == 3, "the weighted split exchanges 3-byte planes"
            words = weighted_helper_words(width, height, world, self.root_run)
            period = self.root_run + world - 1
            mine = pattern_tiles(height, 0, period, self.root_run) if rank == 0 else pattern_tiles(height, self.root_run + rank - 1, period, 1)
            self.local_rows = len(mine) * TILE_ROWS
        if self.solo:
            self.local_rows = tile_count(height) * TILE_ROWS if rank == 0 else 0
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
                if not (self.solo and rank != 0):
                    s.framebuffer = torch.zeros(width * height * 4, dtype=torch.int32, device=dev)
                    r.set_output(s.framebuffer.data_ptr())
            elif self.weighted:
                s.plane3 = torch.zeros(words * 3, dtype=torch.uint8, device=dev)       # send buffer (the root's is never read)
                if rank == 0:
                    r.set_tile_pattern(0, period, self.root_run, False)
                    s.framebuffer = torch.zeros(width * height * 4, dtype=torch.int32, device=dev)
                    r.set_output(s.framebuffer.data_ptr())
                    s.gathered = torch.zeros((world, words * 3), dtype=torch.uint8, device=dev)
                else:
                    r.set_tile_pattern(self.root_run + rank - 1, period, 1, True)
                    s.plane = torch.zeros(words, dtype=torch.int32, device=dev)
                    r.set_plane_output(s.plane.data_ptr())
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
            if not self.weighted:
                self._root_fb = torch.zeros(width * height * 4, dtype=torch.int32, device=dev)
            self.side = torch.cuda.Stream(device=dev) if (pipeline and self.on_gpu) else None
        if self.on_gpu:
            torch.cuda.synchronize(dev)     # the zero fills above ran on torch's stream; the slots launch on their own

    @property
    def framebuffer(self):
        """Device tensor holding the most recently submitted frame (16 B/pixel); complete after a device sync."""
        if self.exchange and not self.weighted:
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
            if self.solo and self.rank != 0:
                return                        # rank 0 renders the whole frame
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
        if self.weighted:
            self._exchange_weighted(slot)
            return
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

    def _exchange_weighted(self, slot):
        """Helpers: pack and send their plane.  Root: its own tiles are already in the slot's framebuffer; receive the
        helpers' planes and write their tiles into it."""
        import torch
        import torch.distributed as td
        stream = slot.stream.cuda_stream if self.on_gpu else None
        if self.rank != 0:
            slot.r.pack_colour_plane3(slot.plane.data_ptr(), slot.plane3.data_ptr(), slot.plane.numel(), stream=stream)
        glist = list(slot.gathered.unbind(0)) if self.rank == 0 else None
        work = td.gather(slot.plane3, glist, dst=0, async_op=True)      # the one exchange step of the frame
        slot.work = work
        if self.rank == 0:
            args = (slot.gathered.data_ptr(), slot.framebuffer.data_ptr(), self.W, self.H, self.world, self.root_run, slot.gathered.shape[1])
            if self.side is None:
                work.wait()
                slot.r.scatter_helper_planes3(*args)
            else:
                with torch.cuda.stream(self.side):
                    work.wait()
                    slot.r.scatter_helper_planes3(*args, stream=self.side.cuda_stream)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    slot.scattered = ev
        elif not self.pipeline:
            work.wait()
