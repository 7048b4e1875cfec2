"""The N>1 frame path on CPU: row-tile sharding + ONE gather + root-side reassembly, world_size 2 and 3,
gloo backend.  The render itself cannot run without a GPU (no CPU fallback in the product), so each rank's
colour plane is cut from one oracle frame; what is exercised is every piece of host logic the multi-GPU
bench relies on: tile arithmetic, padded plane sizes, the gather call pattern and the reassembly.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

import oracle_ffi
from conftest import load_config
from relativitypathtracer_amd import dist as rdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_arithmetic():
    for H in (1, 7, 8, 9, 72, 77, 2160, 4320):
        for world in (1, 2, 3, 8):
            rows = []
            for r in range(world):
                tr = rdist.tile_rows_of_rank(H, r, world)
                assert len(tr) == rdist.local_tile_count(H, r, world) <= rdist.max_local_tiles(H, world)
                rows += [y for rg in tr for y in rg]
            assert sorted(rows) == list(range(H))          # every row rendered exactly once
    assert rdist.plane_words(3840, 2160, 8) == 34 * 8 * 3840      # 270 tiles / 8 ranks -> 34 padded


def test_reassemble_is_inverse_of_extract():
    rng = np.random.default_rng(1)
    for (W, H, world) in [(33, 77, 3), (64, 72, 2), (16, 8, 8), (40, 100, 1)]:
        frame = rng.integers(0, 2 ** 32, size=(H, W), dtype=np.uint32)
        planes = np.stack([rdist.extract_plane(frame, W, H, r, world) for r in range(world)])
        out = rdist.reassemble_planes(planes, W, H, world)
        assert np.array_equal(out["rgba"].view(np.uint32).reshape(H, W), frame)
        assert np.array_equal(out["x"].reshape(H, W)[5 % H], np.arange(W, dtype=np.float32))
        assert np.array_equal(out["y"].reshape(H, W)[:, 0], np.arange(H, dtype=np.float32))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, frame_path, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frame = np.load(frame_path)
        plane = torch.from_numpy(rdist.extract_plane(frame, W, H, rank, world).view(np.int32).copy())
        assert plane.numel() == rdist.plane_words(W, H, world)
        gathered = torch.zeros((world, plane.numel()), dtype=torch.int32) if rank == 0 else None
        td.gather(plane, list(gathered.unbind(0)) if rank == 0 else None, dst=0)     # the one exchange step
        if rank == 0:
            out = rdist.reassemble_planes(gathered.numpy().view(np.uint32), W, H, world)
            np.save(out_path, out["rgba"].view(np.uint32).reshape(H, W))
        td.barrier()
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_frame(tmp_path, world):
    W, H = 96, 77          # 10 tiles, the last one partial
    scene = load_config("shadows")
    px, _, _ = oracle_ffi.render(scene, W, H, want_rgb=False)
    frame = px["rgba"].view(np.uint32).reshape(H, W)
    fp, op = str(tmp_path / "frame.npy"), str(tmp_path / "out.npy")
    np.save(fp, frame)
    mp.spawn(_worker, args=(world, _free_port(), W, H, fp, op), nprocs=world, join=True)
    assert np.array_equal(np.load(op), frame)


def _rank_rows(ptr, ctype, world, stride_elems, row_elems):
    """View of `world` rows of row_elems elements that lie stride_elems apart, starting at ptr (no read past the last row)."""
    import ctypes as C
    flat = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=((world - 1) * stride_elems + row_elems,))
    return np.lib.stride_tricks.as_strided(flat, shape=(world, row_elems), strides=(stride_elems * flat.itemsize, flat.itemsize))


class _OracleSlot:
    """Stand-in for one frame slot's context on a machine without a GPU: the same calls dist.FrameSharder makes
    on a Renderer, answered by the CPU oracle and numpy (test infrastructure, like the oracle itself)."""

    def __init__(self, scene, W, H):
        self.scene, self.W, self.H = scene, W, H
        self.objects = None
        self.rows, self.run = (0, 1, False), 1
        self.plane_ptr = self.out_ptr = 0

    def set_rows(self, first, step, plane):
        self.rows, self.run = (first, step, plane), 1

    def set_tile_pattern(self, first, step, run, plane):
        self.rows, self.run = (first, step, plane), run

    def set_output(self, ptr):
        self.out_ptr = ptr or 0

    def sync(self):
        pass

    def set_plane_output(self, ptr):
        self.plane_ptr = ptr

    def set_objects(self, objects_bytes):
        self.objects = np.array(objects_bytes, dtype=np.uint8, copy=True)

    def render_async(self):
        import ctypes as C
        first, step, plane = self.rows
        px, _, _ = oracle_ffi.render(self.scene, self.W, self.H, want_rgb=False, objects=self.objects)
        tiles = rdist.pattern_tiles(self.H, first, step, self.run)
        if plane:                                   # local tile order, 4 B/px
            assert self.plane_ptr
            frame = px["rgba"].view(np.uint32).reshape(self.H, self.W)
            mine = np.zeros((len(tiles) * 8, self.W), dtype=np.uint32)
            for k, t in enumerate(tiles):
                rows = frame[t * 8:min(t * 8 + 8, self.H)]
                mine[k * 8:k * 8 + len(rows)] = rows
            C.memmove(self.plane_ptr, mine.ctypes.data, mine.nbytes)
        elif self.out_ptr:                          # its tiles of the 16 B/px framebuffer, in place
            fb = np.ctypeslib.as_array(C.cast(self.out_ptr, C.POINTER(C.c_uint8)), shape=(self.H, self.W * 16))
            src = px.view(np.uint8).reshape(self.H, self.W * 16)
            for t in tiles:
                fb[t * 8:min(t * 8 + 8, self.H)] = src[t * 8:min(t * 8 + 8, self.H)]

    def scatter_colour_plane(self, planes_ptr, out_ptr, W, H, world, stride_words, stream=None):
        import ctypes as C
        planes = _rank_rows(planes_ptr, C.c_uint32, world, stride_words, rdist.plane_words(W, H, world))
        out = rdist.reassemble_planes(np.ascontiguousarray(planes), W, H, world)
        C.memmove(out_ptr, out.ctypes.data, out.nbytes)

    def scatter_helper_planes3(self, planes3_ptr, out_ptr, W, H, world, root_run, stride_bytes, stream=None):
        import ctypes as C
        from relativitypathtracer_amd.renderer import PIXEL_DTYPE
        raw = _rank_rows(planes3_ptr, C.c_uint8, world, stride_bytes, 3 * rdist.weighted_helper_words(W, H, world, root_run))
        fb = np.ctypeslib.as_array(C.cast(out_ptr, C.POINTER(C.c_uint8)), shape=(H * W * 16,)).view(PIXEL_DTYPE).reshape(H, W)
        period = root_run + world - 1
        for j in range(1, world):
            words = rdist.unpack_plane3(raw[j]).reshape(-1, W)
            for k, t in enumerate(rdist.pattern_tiles(H, root_run + j - 1, period, 1)):
                for y in range(t * 8, min(t * 8 + 8, H)):
                    fb["x"][y] = np.arange(W, dtype=np.float32)
                    fb["y"][y] = np.float32(y)
                    fb["rgba"][y] = words[k * 8 + y - t * 8].view(np.uint8).reshape(W, 4)

    def pack_colour_plane3(self, plane4_ptr, plane3_ptr, pixels, stream=None):
        import ctypes as C
        words = np.ctypeslib.as_array(C.cast(plane4_ptr, C.POINTER(C.c_uint32)), shape=(pixels,))
        packed = rdist.pack_plane3(words)
        C.memmove(plane3_ptr, packed.ctypes.data, packed.nbytes)

    def scatter_colour_plane3(self, planes3_ptr, out_ptr, W, H, world, stride_bytes, stream=None):
        import ctypes as C
        raw = _rank_rows(planes3_ptr, C.c_uint8, world, stride_bytes, 3 * rdist.plane_words(W, H, world))
        planes = np.stack([rdist.unpack_plane3(raw[r]) for r in range(world)])
        out = rdist.reassemble_planes(planes, W, H, world)
        C.memmove(out_ptr, out.ctypes.data, out.nbytes)


def _sharder_worker(rank, world, port, W, H, frames, snap_path, out_path, plane_bytes, root_run=None, group=3, mid_flush=-1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from relativitypathtracer_amd import Scene
        scene = Scene.from_file("shadows")
        snaps = np.load(snap_path)                      # Object[] of every frame, computed once by the test
        slots = [_OracleSlot(scene, W, H) for _ in range(3)]
        if root_run == "auto":                       # measure + agree, as bench.py does on a real node
            model_run, info = rdist.calibrate_split(slots, snaps[0], W, H, rank, world, device="cpu", frames=3)
            assert model_run in (0, 1, 2, 4, 8, 16) and info["frame_ms_one_rank"] > 0
            root_run, tried = rdist.autotune_split(slots, snaps[0], W, H, rank, world, sorted({0, 2, model_run}), device="cpu",
                                                   frames_per_exchange=group, rounds=1)
            assert root_run in tried and all(v > 0 for v in tried.values())
        sharder = rdist.FrameSharder(slots, W, H, rank, world, device="cpu", plane_bytes=plane_bytes, root_run=root_run,
                                     frames_per_exchange=group)
        assert sharder.depth == 3 and (sharder.exchange or sharder.solo)
        seen = {}

        def collect(first, count):                      # the frames of a batch that has just been exchanged
            if rank != 0:
                return
            for f in range(first, first + count):
                fb = sharder.slots[f % 3].framebuffer if sharder.solo else sharder.batches[(f // sharder.group) % 2].fbs[f % sharder.group]
                seen[f] = fb.numpy()[:H * W * 4].view(np.uint8).reshape(H * W, 16)[:, 8:12].copy().view(np.uint32)[:, 0]   # (full16 pads to whole tiles)

        for f in range(frames):
            sharder.render_and_gather(snaps[f])
            if sharder.solo:
                collect(f, 1)
            elif (f + 1) % sharder.group == 0:          # without streams the root reassembles as soon as the batch is complete
                collect(f + 1 - sharder.group, sharder.group)
            elif f == mid_flush:                        # a flush in mid-batch (bench.py: between warm-up and the timed region)
                sharder.flush()
                collect(f - f % sharder.group, f % sharder.group + 1)
        sharder.flush()
        if not sharder.solo and frames % sharder.group:
            collect(frames - frames % sharder.group, frames % sharder.group)
        td.barrier()
        if rank == 0:
            assert sharder.framebuffer is not None
            last = sharder.framebuffer.numpy().view(np.uint8).reshape(H * W, 16)[:, 8:12].copy().view(np.uint32)[:, 0]
            assert np.array_equal(last, seen[frames - 1])
            np.save(out_path, np.stack([seen[f] for f in range(frames)]))
    finally:
        td.destroy_process_group()


def test_three_byte_plane_round_trip():
    rng = np.random.default_rng(7)
    words = (rng.integers(0, 2 ** 24, size=4096, dtype=np.uint32) | np.uint32(1 << 24))
    packed = rdist.pack_plane3(words)
    assert packed.dtype == np.uint8 and packed.size == 3 * words.size
    assert np.array_equal(rdist.unpack_plane3(packed), words)


@pytest.mark.parametrize("world,plane_bytes,root_run,group", [(2, 3, None, 3), (3, 3, None, 1), (2, 4, None, 2), (2, 3, 4, 4), (3, 3, 2, 3),
                                                              (3, 3, 1, 2), (2, 3, 0, 3), (2, 3, "auto", 3), (2, 3, 2, 4), (2, 16, None, 3), (3, 16, None, 2)])
def test_frame_sharder_three_frames_in_flight(tmp_path, world, plane_bytes, root_run, group):
    """dist.FrameSharder itself, world 2 and 3 over gloo: three frame slots rotating over seven different frames
    (camera clock running), one gather per `group` frames (the last batch partial, sent by flush()) — of 3-byte
    planes (the default), of the 4-byte planes as rendered, or of whole 16-byte pixels (the naive exchange, plane_bytes 16);
    with the equal split (root_run None), the weighted split (rank 0 renders root_run of every root_run + N - 1 tiles in
    place), rank 0 alone (0), and the split measured and agreed on by calibrate_split ("auto").  Rank 0's framebuffer
    after every frame must be that frame."""
    from relativitypathtracer_amd import Scene
    W, H, frames = 64, 77, 7
    scene = Scene.from_file("shadows")
    scene.set_paused(False)
    scene.set_camera((0, 0, 0), 14.0)
    snaps, want = [], []
    for f in range(frames):
        scene.advance_time(400)
        scene.update_objects()
        snaps.append(scene.buffers()["objects"].copy())
        px, _, _ = oracle_ffi.render(scene, W, H, want_rgb=False)
        want.append(px["rgba"].view(np.uint32).reshape(H * W).copy())
    assert any(not np.array_equal(want[0], w) for w in want[1:])
    sp, op = str(tmp_path / "snaps.npy"), str(tmp_path / "out.npy")
    np.save(sp, np.stack(snaps))
    mp.spawn(_sharder_worker, args=(world, _free_port(), W, H, frames, sp, op, plane_bytes, root_run, group, 4 if (root_run, group) == (2, 4) else -1),
             nprocs=world, join=True)
    got = np.load(op)
    for f in range(frames):
        assert np.array_equal(got[f], want[f]), f"frame {f}"
