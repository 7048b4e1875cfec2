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
