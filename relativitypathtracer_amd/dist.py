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
                    overhead_s: float = 0.0, frames_per_exchange: int = 1) -> int:
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
        # one gather carries frames_per_exchange planes: its fixed cost is shared, its bytes are not
        wire = gather_base_s / max(frames_per_exchange, 1) + gather_s_per_byte * 3 * weighted_helper_words(width, height, world, run)
        # the root also reassembles the helpers' tiles (3 B read + 16 B written per pixel, at a conservative 3 TB/s)
        reassembly = 19.0 * width * height * (1.0 - share_root) / 3.0e12
        t = max(frame_s * share_root + reassembly + overhead_s, frame_s * share_helper + overhead_s, wire)
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


def calibrate_split(renderers, objects, width: int, height: int, rank: int, world: int, device=None, frames: int = 30,
                    frames_per_exchange: int = 1, comm=None):
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
        if comm is not None:            # the exchange the frames will use: ncclGather through ctypes on the current stream
            from . import rccl
            cur = torch.cuda.current_stream(dev).cuda_stream

            def one():
                comm.gather(send.data_ptr(), recv.data_ptr() if rank == 0 else 0, nbytes, rccl.NCCL_UINT8, 0, cur)
            for _ in range(2):
                one()
            sync()
            td.barrier()
            sync()
            t1 = time.perf_counter()
            for _ in range(reps):
                one()
            sync()
            return (time.perf_counter() - t1) / reps
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
        choice[0] = choose_root_run(frame_s, base, per_byte, width, height, world, frames_per_exchange=frames_per_exchange)
    td.broadcast(choice, src=0)
    sync()
    return int(choice[0]), {"frame_ms_one_rank": round(frame_s * 1e3, 4), "gather_base_ms": round(base * 1e3, 4),
                            "gather_GBps_per_rank": round(1e-9 / per_byte, 2) if per_byte > 0 else None}


def autotune_split(renderers, objects, width: int, height: int, rank: int, world: int, candidates, device=None,
                   frames_per_exchange: int = 1, force_gather: bool = False, rounds: int = 3, comm=None):
    """Run each candidate arrangement (values of FrameSharder's root_run: None = the equal split, 0 = rank 0 alone, a power of two = weighted)
    for `rounds` batches of frames and keep the fastest.  A candidate's time is the MAX over ranks of the wall time
    per frame (one all_reduce), so every rank holds the same numbers and picks the same arrangement; ties go to the
    smaller root_run.  Collectives per candidate: what its frames need, one barrier, one all_reduce — the same on
    every rank.  Returns (root_run, {candidate: seconds per frame})."""
    import time
    import torch
    import torch.distributed as td
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    results = {}
    for cand in candidates:
        sharder = FrameSharder(renderers, width, height, rank, world, force_gather=force_gather, device=device, root_run=cand,
                               frames_per_exchange=frames_per_exchange, comm=comm)
        n = rounds * max(sharder.group, 1)
        for _ in range(max(sharder.group, 1)):
            sharder.render_and_gather(objects)
        sharder.flush()
        sync()
        td.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            sharder.render_and_gather(objects)
        sharder.flush()
        for r in renderers:
            r.sync()
        sync()
        t = torch.tensor([(time.perf_counter() - t0) / n], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        results[cand] = float(t[0])
        del sharder
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    best = min(results, key=lambda c: (results[c], -1 if c is None else c))
    return best, results


class _Slot:
    """One frame being rendered: a context (rpt_ctx) on its own stream with its render target."""
    __slots__ = ("r", "stream", "framebuffer", "plane", "frames")


class _Batch:
    """The frames of one exchange: send / receive buffers, the root's framebuffers, and the events that order them."""
    __slots__ = ("send", "recv", "fbs", "ready", "work", "done", "gathered", "count")


class FrameSharder:
    """Owns the output tensors of one rank and runs render (+ exchange + reassembly) frame after frame.

    Frames in flight.  A frame's critical path is the serial octree walk of its dearest pixel, so
    one frame alone leaves most of the GPU idle for most of its duration.  The sharder therefore keeps
    ``len(renderers)`` frames in flight: every slot is a context of its own (scene resident once per context) on
    its own stream with its own render target, frame f runs in slot ``f mod depth``, and kernels of consecutive
    frames overlap on the device.  Each frame still gets its own ``rpt_set_objects`` (the host may change
    ``Object[]`` between any two frames) and is rendered completely; ``framebuffer`` is the last submitted frame.

    With more than one rank the exchange is ONE gather (RCCL over xGMI) per `frames_per_exchange` frames: a
    collective costs tens of microseconds of host and launch time however small it is — as much as a frame — so
    the planes of consecutive frames travel together.  Rendering, the exchange and the root's reassembly run on
    different queues (the slots' streams, an exchange stream, a side stream) over two alternating batches of
    buffers; the host never blocks (buffer reuse is ordered by events).  Every rank issues the same collectives in
    the same order; ``flush()`` sends a partial last batch and must be called (by every rank) before the
    framebuffer is read or a timed region ends.
    """

    def __init__(self, renderers, width: int, height: int, rank: int, world: int, force_gather: bool = False,
                 pipeline: bool = True, device=None, plane_bytes: int = 3, root_run: Optional[int] = None,
                 frames_per_exchange: int = 1, comm=None):
        """`device`: where the output tensors live; default the current GPU.  A CPU device (tests/test_dist_gloo.py:
        gloo, stand-in renderers) runs the same rotation and exchange without streams.
        `plane_bytes`: bytes per pixel on the wire — 3 (default: the alpha byte of a packed colour is the constant
        1, so a small kernel drops it before the gather), 4 (the rendered plane as it is), or 16: the NAIVE exchange
        SURVEY.md §8(e) asks to keep for comparison (`--gather=full16`) — every rank renders its tiles as whole 16-byte
        pixels, copies them together, the gather carries 16 B/px (x and y of every pixel included, though they never
        change), and the root copies each rank's tiles to their rows.  Equal split only.
        `root_run`: None = equal interleaved split (tile k -> rank k mod N, every plane gathered).  A power of two =
        the WEIGHTED split: per period of root_run + N - 1 tiles rank 0 renders root_run tiles straight into its
        framebuffer and rank j the single tile root_run + j - 1 into a 3 B/px plane — pixels rendered where they are
        needed cross no link, so the root takes the larger share (choose_root_run sizes it).  0 = rank 0 renders the
        whole frame and nothing is exchanged (the other ranks idle): the arrangement to fall back to when the
        exchange is slower than rendering.
        `pipeline` False: one slot, one frame per exchange, every stage waited for.
        `comm`: an rccl.Communicator — the gather is then ONE ncclGather enqueued through ctypes on the exchange stream (a few
        microseconds of host time; torch.distributed is only the rendezvous), and one frame per exchange is the default: no
        display latency is traded for host time.  None: torch.distributed.gather (gloo tests, the CPU rehearsals)."""
        self.comm = comm
        import torch
        if not isinstance(renderers, (list, tuple)):
            renderers = [renderers]
        if not pipeline:
            renderers, frames_per_exchange = renderers[:1], 1
        self.W, self.H, self.rank, self.world = width, height, rank, world
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dev, self.on_gpu = dev, dev.type == "cuda"
        self.exchange = world > 1 or force_gather      # force_gather: run the plane/gather/scatter path with one rank
        self.root_run = root_run if ((world > 1 or force_gather) and root_run is not None) else None
        self.solo = self.root_run == 0                  # rank 0 renders everything, no exchange
        self.weighted = bool(self.root_run)
        if self.solo:
            self.exchange = False
        assert plane_bytes in (3, 4, 16) and not (self.weighted and plane_bytes != 3), "the weighted split exchanges 3-byte planes"
        self.full16 = self.exchange and plane_bytes == 16
        self.plane_bytes, self.pipeline, self.depth = plane_bytes, pipeline, len(renderers)
        self.group = max(1, int(frames_per_exchange)) if self.exchange else 1
        self.frame, self.last, self.last_fb = 0, None, None

        # which tiles this rank renders, how large a plane is
        if self.weighted:
            period = self.root_run + world - 1
            self.words = weighted_helper_words(width, height, world, self.root_run)
            mine = pattern_tiles(height, 0, period, self.root_run) if rank == 0 else pattern_tiles(height, self.root_run + rank - 1, period, 1)
            self.local_rows = len(mine) * TILE_ROWS
        else:
            self.words = plane_words(width, height, world)
            self.local_rows = local_tile_count(height, rank, world) * TILE_ROWS
        if self.solo:
            self.local_rows = tile_count(height) * TILE_ROWS if rank == 0 else 0
        self.plane_unit = self.words * {3: 3, 4: 1, 16: 4}[plane_bytes]               # elements of one frame's plane on the wire
        wire_dtype = torch.uint8 if plane_bytes == 3 else torch.int32
        renders_plane = self.exchange and not (self.weighted and rank == 0) and not self.full16   # the weighted root renders in place
        self.tile_words = TILE_ROWS * width * 4                                        # int32 words of one tile of 16-byte pixels
        padded_fb = tile_count(height) * self.tile_words                               # a framebuffer of whole tiles (full16 views it by tile)

        self.slots = []
        for r in renderers:
            s = _Slot()
            s.r, s.frames, s.framebuffer, s.plane = r, 0, None, None
            # The render kernels and the tensors they touch share ONE real stream per slot.  torch's default stream has
            # handle 0, which the C-ABI reads as "the context's own stream": never it.
            s.stream = torch.cuda.Stream(device=dev) if self.on_gpu else None
            if self.on_gpu:
                r.set_stream(s.stream.cuda_stream)
            if not self.exchange:
                r.set_rows(0, 1, False)
                if not (self.solo and rank != 0):
                    s.framebuffer = torch.zeros(width * height * 4, dtype=torch.int32, device=dev)
                    r.set_output(s.framebuffer.data_ptr())
            elif self.weighted:
                if rank == 0:
                    r.set_tile_pattern(0, period, self.root_run, False)
                else:
                    r.set_tile_pattern(self.root_run + rank - 1, period, 1, True)
            elif self.full16:                     # its tiles as whole pixels, where they lie in a framebuffer of its own
                r.set_rows(rank, world, False)
                s.framebuffer = torch.zeros(padded_fb, dtype=torch.int32, device=dev)
                r.set_output(s.framebuffer.data_ptr())
            else:
                r.set_rows(rank, world, True)
            if renders_plane:
                s.plane = torch.zeros(self.words, dtype=torch.int32, device=dev)
                r.set_plane_output(s.plane.data_ptr())
            self.slots.append(s)
        self.r = self.slots[0].r

        self.batches = []
        self.xstream = self.side = None
        if self.exchange:
            for _ in range(2 if pipeline else 1):
                b = _Batch()
                b.send = torch.zeros(self.group * self.plane_unit, dtype=wire_dtype, device=dev)
                b.recv = torch.zeros((world, self.group * self.plane_unit), dtype=wire_dtype, device=dev) if rank == 0 else None
                b.fbs = [torch.zeros(padded_fb if self.full16 else width * height * 4, dtype=torch.int32, device=dev)
                         for _ in range(self.group)] if rank == 0 else None
                b.ready = [torch.cuda.Event() for _ in range(self.group)] if self.on_gpu else None
                b.done = torch.cuda.Event() if self.on_gpu else None
                b.gathered = torch.cuda.Event() if self.on_gpu else None
                b.work, b.count = None, 0
                self.batches.append(b)
            if self.on_gpu:
                self.xstream = torch.cuda.Stream(device=dev)
                self.side = torch.cuda.Stream(device=dev) if rank == 0 else None
        if self.on_gpu:
            torch.cuda.synchronize(dev)     # the zero fills above ran on torch's stream; the slots launch on their own

    @property
    def framebuffer(self):
        """Tensor holding the most recently submitted frame (16 B/pixel) on rank 0; complete after flush() + a device sync."""
        if self.exchange:
            return self.last_fb[:self.W * self.H * 4] if (self.full16 and self.last_fb is not None) else self.last_fb
        return (self.last or self.slots[0]).framebuffer

    def render_and_gather(self, objects=None):
        """Submit one frame: Object[] refresh (if given: a Scene or raw bytes), render, and with N > 1 its part of the
        exchange (the gather itself goes out with the last frame of a batch, or with flush())."""
        slot = self.slots[self.frame % self.depth]
        f = self.frame
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
        batch = self.batches[(f // self.group) % len(self.batches)]
        k = f % self.group                     # position of the frame in its batch
        stream = slot.stream.cuda_stream if self.on_gpu else None
        if self.on_gpu and batch.work is not None:
            slot.stream.wait_event(batch.done)             # the batch's buffers are free again (exchange and reassembly over)
        if self.rank == 0:
            self.last_fb = batch.fbs[k]
        if objects is not None:
            slot.r.set_objects(objects)
        if self.weighted and self.rank == 0:
            slot.r.set_output(batch.fbs[k].data_ptr())     # the root's own tiles go straight into this frame's framebuffer
        slot.r.render_async()
        if slot.plane is not None:                          # this rank's plane of the frame -> its place in the batch
            unit = batch.send[k * self.plane_unit:(k + 1) * self.plane_unit]
            if self.plane_bytes == 3:
                slot.r.pack_colour_plane3(slot.plane.data_ptr(), unit.data_ptr(), self.words, stream=stream)
            elif self.on_gpu:
                import torch
                with torch.cuda.stream(slot.stream):
                    unit.copy_(slot.plane, non_blocking=True)
            else:
                unit.copy_(slot.plane)
        if self.full16:                                     # this rank's tiles, 16 B/px, copied together
            import torch
            n_local = local_tile_count(self.H, self.rank, self.world)
            unit = batch.send[k * self.plane_unit:(k + 1) * self.plane_unit].view(-1, self.tile_words)
            mine = slot.framebuffer.view(-1, self.tile_words)[self.rank::self.world]
            if self.on_gpu:
                with torch.cuda.stream(slot.stream):
                    unit[:n_local].copy_(mine, non_blocking=True)
            else:
                unit[:n_local].copy_(mine)
        if self.on_gpu:
            batch.ready[k].record(slot.stream)
        batch.count = k + 1
        if batch.count == self.group:
            self._exchange(batch)

    def flush(self):
        """Send a partial last batch.  Every rank must call it at the same point (it may issue a collective)."""
        if not self.exchange or self.frame == 0:
            return
        batch = self.batches[((self.frame - 1) // self.group) % len(self.batches)]
        if 0 < batch.count < self.group:
            self._exchange(batch)

    def _exchange(self, batch):
        """ONE gather for the batch's frames, then (rank 0) the reassembly of each of them."""
        import torch
        import torch.distributed as td
        n = batch.count
        batch.count = 0
        if self.on_gpu and self.comm is not None:          # ONE ncclGather through ctypes: no torch object on the per-frame path
            from . import rccl
            for k in range(n):
                self.xstream.wait_event(batch.ready[k])
            self.comm.gather(batch.send.data_ptr(), batch.recv.data_ptr() if self.rank == 0 else 0, batch.send.numel(),
                             rccl.NCCL_UINT8 if self.plane_bytes == 3 else rccl.NCCL_INT32, 0, self.xstream.cuda_stream)
            batch.work = True
            if self.rank != 0:
                batch.done.record(self.xstream)
                return
            batch.gathered.record(self.xstream)
            self.side.wait_event(batch.gathered)
            if self.full16:                                 # (its reassembly is torch copies: they need torch's current stream)
                with torch.cuda.stream(self.side):
                    self._reassemble(batch, n)
            else:
                self._reassemble(batch, n)                  # the scatter kernels take their stream as an argument
            batch.done.record(self.side)
            if not self.pipeline:
                self.side.synchronize()
            return
        glist = list(batch.recv.unbind(0)) if self.rank == 0 else None
        if self.on_gpu:
            for k in range(n):
                self.xstream.wait_event(batch.ready[k])
            with torch.cuda.stream(self.xstream):
                work = td.gather(batch.send, glist, dst=0, async_op=True)      # the exchange step of these frames
                if self.rank != 0:
                    work.wait()
                    batch.done.record(self.xstream)
        else:
            work = td.gather(batch.send, glist, dst=0, async_op=True)
            work.wait()
        batch.work = work
        if self.rank != 0:
            return
        if self.on_gpu:
            with torch.cuda.stream(self.side):
                work.wait()
                self._reassemble(batch, n)
                batch.done.record(self.side)
            if not self.pipeline:
                self.side.synchronize()
        else:
            self._reassemble(batch, n)

    def _reassemble(self, batch, n):
        """rank 0: the gathered planes of the batch's n frames -> their framebuffers (on the side stream when on the GPU)"""
        side = self.side.cuda_stream if self.on_gpu else None
        for k in range(n):
            src = batch.recv[:, k * self.plane_unit:(k + 1) * self.plane_unit]
            # the k-th frame's planes lie (group * plane_unit) elements apart, one per rank
            stride = batch.recv.shape[1] * batch.recv.element_size()
            if self.full16:                # every rank's tiles back to their rows: N strided copies of whole pixels
                fb = batch.fbs[k].view(-1, self.tile_words)
                for j in range(self.world):
                    fb[j::self.world].copy_(src[j].view(-1, self.tile_words)[:local_tile_count(self.H, j, self.world)], non_blocking=True)
                continue
            if self.weighted:
                self.r.scatter_helper_planes3(src.data_ptr(), batch.fbs[k].data_ptr(), self.W, self.H, self.world, self.root_run,
                                              stride, stream=side)
            elif self.plane_bytes == 3:
                self.r.scatter_colour_plane3(src.data_ptr(), batch.fbs[k].data_ptr(), self.W, self.H, self.world, stride, stream=side)
            else:
                self.r.scatter_colour_plane(src.data_ptr(), batch.fbs[k].data_ptr(), self.W, self.H, self.world, stride // 4, stream=side)
