"""RCCL through ctypes: the per-frame exchange without torch.distributed in its way.

``torch.distributed.gather`` costs 30+ microseconds of host time per call and 90 of bookkeeping per exchanged frame
(profiles/r02_host_cost.txt, r03_exchange_host_cost.txt) — as much as a frame takes to render — which is why round 2 sent four
frames per gather: three frames of display latency for an interactive renderer.  The native host
(examples/rpt_multi_gpu_main.cpp) calls ncclGather directly and pays a tenth of that.  This module gives the Python harness the
same call: one communicator per process (ncclCommInitRank; the 128-byte unique id travels once over torch.distributed, which
stays the RENDEZVOUS — the driver launches the ranks with torchrun), then one ``ncclGather`` per exchange, enqueued on the
exchange stream by ctypes (a few microseconds).  The library is the librccl.so torch itself has loaded (one RCCL per process).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

NCCL_UINT8, NCCL_INT32 = 1, 2          # ncclDataType_t (rccl.h): ncclInt8 0, ncclUint8 1, ncclInt32 2
_lib = None


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        cands = []
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cands.append(os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so"))
        cands += ["/opt/rocm/lib/librccl.so", "librccl.so"]
        err = None
        for p in cands:
            try:
                _lib = C.CDLL(p)
                break
            except OSError as e:
                err = e
        if _lib is None:
            raise RuntimeError(f"librccl.so not found ({err})")
        _lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        _lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        _lib.ncclGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.ncclCommDestroy.argtypes = [C.c_void_p]
        _lib.ncclGetErrorString.argtypes = [C.c_int]
        _lib.ncclGetErrorString.restype = C.c_char_p
        for f in (_lib.ncclGetUniqueId, _lib.ncclCommInitRank, _lib.ncclGather, _lib.ncclCommDestroy):
            f.restype = C.c_int
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().ncclGetErrorString(rc).decode()}")


class Communicator:
    """One RCCL communicator of `world` ranks.  `exchange_id(bytes_or_None) -> bytes`: how rank 0's 128-byte unique id reaches
    the others (torch.distributed.broadcast in bench.py; a pipe, a file, MPI ... anywhere else)."""

    def __init__(self, rank: int, world: int, exchange_id):
        uid = UniqueId()
        if rank == 0:
            _check(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = exchange_id(C.string_at(C.byref(uid), 128) if rank == 0 else None)    # (bytes(c_char array) would stop at the first NUL)
        C.memmove(C.byref(uid), raw, 128)
        self.comm = C.c_void_p()
        _check(lib().ncclCommInitRank(C.byref(self.comm), int(world), uid, int(rank)), "ncclCommInitRank")
        self.rank, self.world = rank, world

    def gather(self, send_ptr: int, recv_ptr: int, count: int, dtype: int, root: int, stream: int):
        """ncclGather (rccl.h:745): rank r's `count` elements land at recv + r * count on the root; enqueued on `stream`."""
        _check(lib().ncclGather(C.c_void_p(send_ptr), C.c_void_p(recv_ptr or 0), C.c_size_t(count), dtype, root, self.comm, C.c_void_p(stream)), "ncclGather")

    def destroy(self):
        if self.comm:
            lib().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def torch_broadcast_id(rank: int, device):
    """exchange_id over an initialised torch.distributed process group (any backend)."""
    def exchange(raw):
        import torch
        import torch.distributed as td
        t = torch.zeros(128, dtype=torch.uint8, device=device if td.get_backend() == "nccl" else "cpu")
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        td.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())
    return exchange
