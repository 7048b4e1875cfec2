#!/usr/bin/env python3
"""Host-side profile of the N > 1 frame path with ONE rank over RCCL (run under torchrun with RPT_FORCE_DIST=1):
where do the microseconds of an exchanged frame go?  usage: torchrun ... tools/exchange_profile.py [native|torch] [frames_per_exchange]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as td
from relativitypathtracer_amd import Scene, dist as rdist, rccl
from relativitypathtracer_amd.renderer import Renderer

mode = sys.argv[1] if len(sys.argv) > 1 else "native"
fpe = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.cuda.set_device(0)
td.init_process_group("nccl", device_id=torch.device("cuda", 0))
W, H = 3840, 2160
scene = Scene.from_file("bunny"); scene.update_objects()
rs = []
for _ in range(3):
    r = Renderer(0)
    if rs: r.share_scene(rs[0])
    else: r.upload_scene(scene)
    r.set_scene_params(scene, W, H)
    rs.append(r)
comm = rccl.Communicator(0, 1, rccl.torch_broadcast_id(0, torch.device("cuda", 0))) if mode == "native" else None
fs = rdist.FrameSharder(rs, W, H, 0, 1, force_gather=True, frames_per_exchange=fpe, comm=comm)
for _ in range(30): fs.render_and_gather(scene)
fs.flush(); torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(N): fs.render_and_gather(scene)
pr.disable()
sub = time.perf_counter() - t0
fs.flush(); torch.cuda.synchronize()
print(f"{mode} fpe={fpe}: host submission {sub / N * 1e6:.1f} us/frame (profiler on), wall {(time.perf_counter() - t0) / N * 1e6:.1f} us/frame")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
td.destroy_process_group()
