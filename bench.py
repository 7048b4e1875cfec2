#!/usr/bin/env python3
"""bench.py — headline benchmark of the render path on MI355X.

Metric (BASELINE.json): Mrays/s (primary rays per second = W*H/frame time/1e6) and ms/frame on
Scenes/bunny.txt at 3840x2160, on N GPUs of one node.

A "step" is one frame: refresh Object[] on the device (rpt_set_objects, as the reference's render()
does every frame, Render.cpp:202) + render every pixel; with N > 1 each rank renders its interleaved
8-row tiles into a 4 B/pixel colour plane, ONE RCCL gather over xGMI brings the planes to rank 0,
and a root-side kernel expands them into the 16 B/pixel framebuffer.  Scene buffers are resident in
HBM before the timed region; the framebuffer stays in device memory (the reference never reads back).

Frames in flight (--inflight, default 4 = one slot per hardware queue of the process): a frame's critical path is the serial octree walk of its dearest pixel,
which leaves most of the GPU idle for most of one frame; consecutive frames are therefore submitted on separate
streams (one context per slot) and overlap on the device.  Every frame is still refreshed, rendered completely
and kept in its slot's framebuffer; `value` is frames/second x pixels over the K timed steps.  The same frames
one at a time (the reference's blocking runKernel) are timed right after and reported as `one_frame_at_a_time`.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (bare: starts the N ranks itself as child processes through the same launcher,
                                         before anything in this process touches the GPU, and relays rank 0's line)
    python bench.py --gpus 2 --dry-run  (launch plumbing only — rendezvous, barriers, the max-over-ranks reduction, the
                                         line's keys — over gloo without touching a device: runs on a box with no GPU)

Rank 0 prints ONE JSON line.  Beside the contract's keys: `kernel_ms` (mean HIP-event duration of a render launch in
the timed region), `one_frame_at_a_time` (the same frames submitted and waited for one by one: latency), `roofline`
(`achieved` from the launch duration and the number of launches sharing the device, `single_launch` without overlap,
`traffic` from the committed rocprofv3 PMC passes), `cpu_baseline` (the oracle on this host's cores), `total_rays`
(primary + shadow rays), `config.sharding` / `config.split_calibration` (N > 1: the arrangement chosen and the
measurements it was chosen from), `check` (with --check: rank 0's framebuffer against the oracle).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the product kernels by variant number (include/rpt.h); which one a launch used is read back from the library (rpt_last_variant)
KERNELS = {
    1: "rpt_render_kernel_v0 (the reference's layouts, no culling)",
    3: "rpt_render_kernel_unculled_w5 (derived layouts, no culling)",
    41: "rpt_render_kernel_ballot_w5 (in-wave cull, natural tile order: what rpt_render_async launches on contexts above 3 Mpx)",
    43: "rpt_render_kernel_ballot_first_w5 (the same with the mesh rows dispatched first and the latency form of the walk: what the blocking rpt_render "
        "launches, and rpt_render_async on contexts of at most RPT_LATENCY_KERNEL_MAX_PIXELS = 3 000 000 pixels)",
    44: "rpt_render_kernel_analytic_w8 (the default kernel without the octree walk compiled in: this workload's Object[] holds no mesh; 8 waves per SIMD)",
}


def kernel_label(variant):
    return KERNELS.get(variant, f"kernel variant {variant} (rpt_set_variant, include/rpt.h)")


HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md): the contract roofline for this path

WORKLOADS = {
    # name: (scene file, camera velocity, camera time)   — BASELINE.md §3
    "bunny": ("bunny", (0.0, 0.0, 0.0), 0.0),
    "shadows": ("shadows", (0.0, 0.0, 0.0), 16.0),
    "arch": ("arch", (0.0, 0.0, 0.95), 5.25),
    "cube": ("cube", (0.0, 0.0, 0.0), 0.0),
    # the reference's other shipped scenes (README.md:96-123), camera states of tests/conftest.py::CONFIGS
    "cubes": ("cubes", (0.3, 0.0, 0.1), 3.0),
    "rulers": ("rulers", (0.0, 0.0, 0.0), 2.5),
    "ladder": ("ladder_paradox", (0.0, 0.0, 0.0), 1.0),
    "soccer": ("soccer", (0.0, 0.0, 0.0), 2.0),
}


def algorithmic_bytes(width, height, n_objects):
    """compulsory bytes of one frame = 16 B/pixel written + Object[] re-read.

    First-touch scene bytes are excluded: they are resident (L2 / Infinity Cache) across frames."""
    return 16 * width * height + 320 * n_objects


def host_cpus():
    """(logical CPUs this process may run on, distinct physical cores among them) — from sched_getaffinity and the
    (physical id, core id) pairs of /proc/cpuinfo; the second is None where /proc/cpuinfo does not say."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        allowed = list(range(os.cpu_count() or 1))
    cores = set()
    try:
        cpu, phys, core = None, None, None
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if line.startswith("processor"):
                    cpu, phys, core = int(line.split(":")[1]), None, None
                elif line.startswith("physical id"):
                    phys = int(line.split(":")[1])
                elif line.startswith("core id"):
                    core = int(line.split(":")[1])
                elif line.strip() == "" and cpu is not None:
                    if cpu in allowed and core is not None:
                        cores.add((phys, core))
                    cpu = None
    except Exception:
        cores = set()
    return len(allowed), (len(cores) or None)


def cpu_quota():
    """CPUs' worth of run time the container's cgroup grants this process (cgroup v2 cpu.max, v1 cfs quota / period), or None
    if unlimited or unknown.  An affinity mask of 256 logical CPUs under a quota of 16 means 16: more threads than that only get
    throttled (measured on the GPU box, round 4: 128 threads 81 Mrays/s, 16 threads 91)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return q / per if q > 0 else None
    except Exception:
        return None


def cpu_baseline(scene, width, height, budget_s=12.0):
    """Time the oracle (the CPU restatement of the reference path) on this host: whole frames of
    the SAME workload, repeated until ~budget_s of wall time, one thread per physical core of this process's affinity mask, plus one
    single-thread frame.  Reported next to the GPU number; it is not the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    affinity, physical = host_cpus()
    quota = cpu_quota()
    # one thread per physical core this process may run on AND is granted run time for (the cgroup's CPU quota caps
    # it: the GPU box shows 256 logical CPUs and grants 16); RPT_CPU_THREADS overrides
    granted = physical or affinity or 1
    if quota:
        granted = max(1, min(granted, int(quota + 0.5)))
    threads = max(1, int(os.environ.get("RPT_CPU_THREADS", granted)))
    oracle_ffi.render(scene, width, min(height, 64), want_rgb=False, threads=threads)   # page in
    frames, t0 = 0, time.perf_counter()
    while True:
        oracle_ffi.render(scene, width, height, want_rgb=False, threads=threads)
        frames += 1
        el = time.perf_counter() - t0
        if el >= budget_s * 0.7 or frames >= 64:
            break
    mt = width * height * frames / el / 1e6
    # single thread on a quarter-height horizontal band through the middle of the frame (the costly rows)
    band = (height * 3 // 8, height * 5 // 8)
    t1 = time.perf_counter()
    oracle_ffi.render(scene, width, height, rows=band, want_rgb=False, threads=1)
    st = width * (band[1] - band[0]) / (time.perf_counter() - t1) / 1e6
    _, _, stats = oracle_ffi.render(scene, width, height, want_rgb=False, want_stats=True, threads=threads)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return {
        "value": round(mt, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "affinity_cpus": affinity, "physical_cores": physical, "cgroup_cpu_quota": quota, "threads_used": threads,
        "sample": f"{frames} whole frame(s) of the same workload ({width}x{height}), {el:.1f} s wall, {threads} threads; "
                  f"1-thread figure on the middle quarter band: {st:.3f} Mrays/s",
        "ms_per_frame": round(el / frames * 1e3, 2), "single_thread_mrays": round(st, 3), "cpu": cpu_model,
        "shadow_rays_per_frame": int(stats["shadow_rays"]),     # counted by the instrumented oracle
    }


def library_sha256():
    """Identity of the running render library: what a PMC summary must have been taken on to be quoted."""
    import hashlib
    so = os.path.join(ROOT, "relativitypathtracer_amd", "librpt_hip.so")
    try:
        with open(so, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def pmc_block(workload, width, height, kernel=None):
    """The block of `kernel` (its full name, e.g. rpt_render_kernel_ballot_first_w5) in a committed rocprofv3 PMC summary
    (tools/profile.sh + tools/pmc_summary.py: separate --pmc passes; one block per kernel) — ONLY if that summary was taken on this
    very build of librpt_hip.so (recorded hash == running hash); otherwise None: counters cannot be collected from inside this
    process, and a number from another build would be stale."""
    import glob
    mine = library_sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_{width}x{height}_pmc_summary.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if not (mine and d.get("build", {}).get("librpt_hip_sha256") == mine):
                continue
            if "kernels" in d:                       # round 4 on: one block per kernel
                blk = d["kernels"].get(kernel)
                if blk is None:
                    continue
                return blk, os.path.relpath(path, ROOT)
            return d, os.path.relpath(path, ROOT)      # (rounds 1-3: kernels blended)
        except Exception:
            continue
    return None, None


def measured_traffic(workload, width, height, kernel=None):
    """HBM bytes per launch of `kernel` from that block: FETCH_SIZE doubled on gfx950, WRITE_SIZE as is."""
    blk, source = pmc_block(workload, width, height, kernel)
    if blk is None:
        return None, None
    return int(sum(v for k, v in blk["derived"].items() if k.startswith("hbm_"))), source


# The other roof of this path.  The kernels are compute: a wave64 vector instruction occupies its SIMD-32 for 2 cycles
# (MI355X_MICROARCH.md: 4 SIMDs per CU, 256 CUs, 2.4 GHz; 157.3 TFLOP/s of fp32 = that rate x 64 lanes x 2 for an FMA).
VALU_PEAK_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 2.0


def measured_valu(workload, width, height, kernel, seconds_per_launch):
    """Vector-ALU issue fraction of `kernel`: SQ_INSTS_VALU per launch (same PMC summary, same hash rule as the traffic) over
    what the chip's 1 024 SIMDs can issue in `seconds_per_launch` — for launches that overlap, the interval between finished
    frames.  lanes_active: of 64, the mean over the kernel's vector instructions (divergence: idle lanes issue all the same)."""
    blk, source = pmc_block(workload, width, height, kernel)
    n = blk and blk.get("SQ_INSTS_VALU", {}).get("mean")
    if not n or not seconds_per_launch:
        return None
    lanes = blk.get("SQ_THREAD_CYCLES_VALU", {}).get("mean")
    return {"wave_instructions_per_launch": int(n), "peak_wave_instructions_per_s": VALU_PEAK_WAVE_INSTR_PER_S,
            "frac": round(n / (VALU_PEAK_WAVE_INSTR_PER_S * seconds_per_launch), 4),
            "lanes_active_of_64": round(lanes / n, 1) if lanes else None, "kernel": kernel, "source": source}


def spread(values):
    """mean / median / min / max of a list of per-frame times (ms)."""
    v = sorted(float(x) for x in values)
    if not v:
        return None
    mid = len(v) // 2
    med = v[mid] if len(v) % 2 else 0.5 * (v[mid - 1] + v[mid])
    return {"mean": round(sum(v) / len(v), 4), "median": round(med, 4), "min": round(v[0], 4), "max": round(v[-1], 4), "frames": len(v)}


def launch_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILD processes through
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and pass rank 0's output through.  Nothing in
    this process has touched the GPU or imported torch.cuda state at this point, and nothing is exec'd."""
    import subprocess
    # The rendezvous belongs to the launcher: its c10d store binds port 0 itself and hands MASTER_ADDR / MASTER_PORT to the ranks.
    # (Rounds 1-3 picked a port here by bind / close / pass-the-number: between the close and the launcher's own bind any other
    # process of the box may take it, and the job dies at start-up — DESIGN.md 8, item 2.)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, RPT_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args):
    """The contract's control flow without a device: one process per rank, gloo rendezvous, barrier-bracketed region,
    max over ranks, rank 0 prints the line (value null, "dry_run": true)."""
    import torch
    import torch.distributed as td
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        td.init_process_group("gloo")
    if world > 1:
        td.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        td.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        td.all_reduce(tt, op=td.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "Mrays/s (primary rays) on Scenes/bunny.txt at 3840x2160", "value": None, "unit": "Mrays/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": f"Scenes/{args.workload}.txt {args.width}x{args.height}"},
                          "comm": {"backend": "gloo" if world > 1 else None, "world_size": world,
                                   "ranks_in_group": td.get_world_size() if world > 1 else 1,
                                   # a real N > 1 run also times BASELINE config 5 in the three arrangements (main(): config5)
                                   "config5": None if world == 1 else {"workload": "Scenes/bunny.txt 7680x4320 (BASELINE config 5)", "ms_per_frame": None,
                                                                       "chosen": None, "dry_run": True}}}), flush=True)
    if world > 1:
        td.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="bunny", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("RPT_FRAMES_IN_FLIGHT", "4")),
                    help="frames in flight (contexts on concurrent streams; default 4 = one per hardware queue of a HIP process — a fifth shares a queue "
                         "and loses, profiles/r03_frames_in_flight.txt); 1 = one frame at a time, as the reference's runKernel()")
    ap.add_argument("--frames-per-exchange", type=int, default=int(os.environ.get("RPT_FRAMES_PER_EXCHANGE", "1")),
                    help="N>1: frames whose planes travel in ONE gather (a collective costs as much host and launch time as a frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="after timing, compare rank 0's framebuffer with the oracle on a few row bands")
    ap.add_argument("--gather", default=os.environ.get("RPT_GATHER", "plane3"), choices=["plane3", "plane4", "full16"],
                    help="what is gathered with N>1: the colour plane at 3 B/pixel (constant alpha byte dropped), at 4 B/pixel as rendered, "
                         "or the naive exchange of whole 16-byte pixels (SURVEY.md 8e: kept to measure against)")
    ap.add_argument("--dry-run", action="store_true", help="launch plumbing only (gloo, no device work): see the module docstring")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))     # bare `python bench.py --gpus N`: this process only launches and relays
    if args.dry_run:
        return dry_run(args)

    import numpy as np
    import torch
    from relativitypathtracer_amd import Scene
    from relativitypathtracer_amd.renderer import Renderer, TILE_ROWS
    from relativitypathtracer_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # one process per GPU: no helper threads for the screen bounds (four polling threads per rank beside the RCCL proxies buy
        # nothing on a still scene; the pool reads this when the first context is created — INTEGRATION.md section 5)
        os.environ.setdefault("RPT_HOST_THREADS", "0")
    n = args.gpus
    n = world                                        # the launcher's word: one rank per GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():      # launcher narrowed the visible devices to one per process
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("RPT_FORCE_DIST") == "1" and "RANK" in os.environ   # rehearse the N>1 path with one rank
    if n > 1 or force_dist:
        import torch.distributed as td
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"       # RCCL's version banner goes to stdout; this program prints ONE line there
        backend = os.environ.get("RPT_BENCH_BACKEND", "nccl")     # "gloo": rehearse N ranks on ONE GPU (RCCL refuses two ranks per device)
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend)

    W, H = args.width, args.height
    scene_name, vel, t = WORKLOADS[args.workload]
    scene = Scene.from_file(scene_name)
    scene.set_camera(vel, t)
    scene.update_objects()
    n_objects = scene.desc().object_count

    # frames in flight: one context per slot, each on its own stream (dist.FrameSharder), one resident scene
    pipeline = os.environ.get("RPT_DIST_PIPELINE", "1") != "0"
    inflight = max(1, args.inflight) if pipeline else 1
    renderers = []
    for _ in range(inflight):
        rr = Renderer(local_rank)
        if renderers:
            rr.share_scene(renderers[0])      # one resident copy of the scene for all frame slots
        else:
            rr.upload_scene(scene)
        rr.set_scene_params(scene, W, H)
        rr.set_variant(args.variant)
        renderers.append(rr)
    r = renderers[0]
    # N > 1: how the frame is split.  auto = measure this node (one rank's frame time, the gather's cost
    # model) and let rank 0 choose between the weighted split, and rendering everything on rank 0 when the exchange
    # would only slow it down; equal = tile k -> rank k mod N; solo / an integer = force that arrangement.
    # The per-frame exchange is ONE ncclGather enqueued through ctypes on the exchange stream (relativitypathtracer_amd/rccl.py);
    # torch.distributed stays the rendezvous (the driver launches the ranks with torchrun) and carries the 128-byte unique id once.
    # RPT_EXCHANGE=torch: torch.distributed.gather instead (what rounds 1-2 used; also what the gloo rehearsals use).
    comm = None
    if (n > 1 or force_dist) and td.get_backend() == "nccl" and os.environ.get("RPT_EXCHANGE", "native") == "native":
        from relativitypathtracer_amd import rccl
        comm = rccl.Communicator(rank, td.get_world_size(), rccl.torch_broadcast_id(rank, torch.device("cuda", local_rank)))
    split, split_info, root_run = os.environ.get("RPT_SPLIT", "auto"), None, None
    split_timings = None
    if (n > 1 or force_dist) and pipeline and args.gather == "plane3" and split != "equal":
        if split == "auto":
            # the cost model proposes a split; it, its neighbours, the EQUAL split and "rank 0 alone" are then tried for a few
            # batches each and the fastest is kept (every rank sees the same max-over-ranks times, so all agree).  All of the
            # timings go into the JSON line: when "rank 0 alone" wins (a 0.1 ms frame is shorter than its own gather), the line
            # still says what the equal and the weighted split cost on this node.
            model_run, split_info = rdist.calibrate_split(renderers, scene, W, H, rank, n, frames_per_exchange=args.frames_per_exchange, comm=comm)
            cands = [None] + sorted({0, model_run} | ({max(1, model_run // 2), min(16, model_run * 2)} if model_run else {4}))
            best, tried = rdist.autotune_split(renderers, scene, W, H, rank, n, cands, frames_per_exchange=args.frames_per_exchange,
                                               force_gather=force_dist, rounds=6, comm=comm)
            name = lambda c: "equal" if c is None else ("solo" if c == 0 else f"weighted_root_run_{c}")
            split_timings = {name(c): round(v * 1e3, 4) for c, v in tried.items()}
            split_info = dict(split_info, model_choice=model_run, tried_ms_per_frame=split_timings, chosen=name(best))
            root_run = best
            if n == 1:
                root_run = None              # one-rank rehearsal: the measurements ran, there is nothing to split
        else:
            root_run = 0 if split == "solo" else int(split)
    frame = rdist.FrameSharder(renderers, W, H, rank, n, force_gather=force_dist, pipeline=pipeline,
                               plane_bytes={"plane3": 3, "plane4": 4, "full16": 16}[args.gather], root_run=root_run,
                               frames_per_exchange=args.frames_per_exchange, comm=comm)   # allocates outputs; N == 1 renders straight into the framebuffers

    animate = os.environ.get("RPT_BENCH_ANIMATE") == "1"     # rehearsal only: every frame differs (camera clock runs)
    clock = [t]

    def step():
        if animate:
            clock[0] += 0.016
            scene.set_camera(vel, clock[0])
            scene.update_objects()
        # per-frame Object[] refresh (as the reference does) + kernel (+ RCCL gather + root scatter when N > 1)
        frame.render_and_gather(scene)

    def barrier():
        frame.flush()                   # a partial last batch of planes goes out now (every rank calls this at the same point)
        torch.cuda.synchronize()
        if n > 1:
            td.barrier()
        torch.cuda.synchronize()

    import ctypes as C

    def end_timing(rr, capacity):
        buf, nfr = (C.c_float * capacity)(), C.c_int()
        rr._check(rr._lib.rpt_timing_end_frames(rr._h, buf, capacity, C.byref(nfr)), "rpt_timing_end_frames")
        return list(buf[:nfr.value])

    batch_ms = []        # per-step time of each fifth of the last timed region (timed() fills it)

    def timed(steps):
        """K steps bracketed by barrier + device sync; returns (wall s, per-launch durations in ms) of this rank."""
        for rr in renderers:
            rr._check(rr._lib.rpt_timing_begin(rr._h, steps), "rpt_timing_begin")
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        submit = time.perf_counter() - t0
        barrier()
        wall = time.perf_counter() - t0
        if rank == 0 and os.environ.get("RPT_BENCH_VERBOSE") == "1":
            print(f"[bench] host submission {submit / steps * 1e3:.4f} ms/step, wall {wall / steps * 1e3:.4f} ms/step", file=sys.stderr)
        # every launch's span on the device's own clock (HIP events; one base event for all frame slots): durations for the roofline, and
        # the region cut into fifths BY COMPLETION TIME for a figure that does not hang on the ramp of a 2-ms region (the host runs
        # ahead of the device — it had submitted all twenty steps of the driver's run before the third frame finished — so host time
        # stamps at submission say nothing)
        launches, ends, first_begin = [], [], None
        for rr in renderers:
            b, e, nfr = (C.c_float * steps)(), (C.c_float * steps)(), C.c_int()
            rr._check(rr._lib.rpt_timing_end_spans(rr._h, renderers[0]._h, b, e, steps, C.byref(nfr)), "rpt_timing_end_spans")
            for k in range(nfr.value):
                launches.append(e[k] - b[k])
                ends.append(e[k])
                first_begin = b[k] if first_begin is None else min(first_begin, b[k])
        ends.sort()
        batch_ms[:] = []
        if len(ends) >= 5:
            cuts = [first_begin] + [ends[(len(ends) * (q + 1)) // 5 - 1] for q in range(5)]
            sizes = [(len(ends) * (q + 1)) // 5 - (len(ends) * q) // 5 for q in range(5)]
            batch_ms[:] = [(cuts[q + 1] - cuts[q]) / max(sizes[q], 1) for q in range(5)]
        return wall, launches

    for _ in range(max(args.warmup, 0)):
        step()
    # The headline workload is a still scene: its Object[] is byte-identical from frame to frame, and rpt_set_objects then reuses
    # the per-object screen bounds instead of recomputing them (a memcmp per object).  An animated scene pays for them on the
    # submitting thread — 1-13 us per object and frame, which for dozens of moving objects is as much as the device needs for the
    # frame (profiles/r02_host_cost.txt).  So the same workload is also timed with the camera clock running (every frame differs),
    # and that figure stands next to the headline one.
    animated = None
    if n == 1 and not force_dist and not animate:
        a_steps = max(10, min(args.steps, 50))
        barrier()
        t0 = time.perf_counter()
        for k in range(a_steps):
            scene.set_camera(vel, t + 0.016 * (k + 1))
            scene.update_objects()
            frame.render_and_gather(scene)
        barrier()
        a_ms = (time.perf_counter() - t0) / a_steps * 1e3
        animated = {"ms_per_step": round(a_ms, 4), "value": round(W * H / a_ms / 1e3, 2), "unit": "Mrays/s", "steps": a_steps,
                    "note": "the same workload with the camera clock advancing 16 ms per frame: every frame's Object[] differs, so the host recomputes "
                            "every object's screen bounds in rpt_set_objects (the still headline frame reuses them)"}
        scene.set_camera(vel, t)
        scene.update_objects()
        for _ in range(frame.depth):      # every slot holds the still frame again
            step()
    elapsed, launches = timed(args.steps)     # launch durations: HIP events on each launch's own stream
    headline_batches = list(batch_ms)
    kernel_ms = sum(launches) / max(len(launches), 1)
    kernel_sum_ms = sum(launches)
    kernel_name = kernel_label(frame.slots[0].r.last_variant())        # what the timed launches were made with

    # the same frames one at a time (submit, wait, submit ...: what the reference's blocking runKernel() does) — the
    # frame LATENCY, and the launch duration without other launches sharing the device
    blocking_ms = blocking_kernel_ms = None
    blocking_frames, blocking_launches = [], []
    if n == 1 and not force_dist:
        nb = max(1, min(args.steps, 50))
        for _ in range(min(max(args.warmup, 0), 3)):     # untimed: the blocking call launches another kernel (43) than the frames in flight
            frame.slots[0].r.set_objects(scene)          # above 3 Mpx did (41), and its first launch in a process is a cold one (0.3-0.4 ms)
            frame.slots[0].r.render()
            frame.slots[0].frames += 1
        barrier()
        r._check(r._lib.rpt_timing_begin(r._h, nb), "rpt_timing_begin")
        for _ in range(nb):
            t0 = time.perf_counter()
            frame.slots[0].r.set_objects(scene)
            frame.slots[0].r.render()
            blocking_frames.append((time.perf_counter() - t0) * 1e3)
        blocking_ms = sum(blocking_frames) / nb
        blocking_launches = end_timing(r, nb)
        blocking_kernel_name = kernel_label(frame.slots[0].r.last_variant())
        if rank == 0 and os.environ.get("RPT_BENCH_VERBOSE") == "1":
            print("[bench] blocking launch durations, ms: " + " ".join(f"{x:.3f}" for x in blocking_launches), file=sys.stderr)
        blocking_kernel_ms = sum(blocking_launches) / max(len(blocking_launches), 1)
        frame.last = frame.slots[0]
        frame.slots[0].frames += nb

    # N > 1: the one shipped workload where sharding can pay — BASELINE config 5, Scenes/bunny.txt at 7680x4320 (0.28 ms per frame on one
    # GPU; the 4K headline frame is shorter than one gather, so its curve is expected to be flat: README.md) — timed in the three
    # arrangements after the headline.  Every rank runs the same collectives.  RPT_BENCH_CONFIG5=0 skips it, =WxH changes its size
    # (rehearsals on one GPU over gloo use a smaller frame).
    config5 = None
    c5 = os.environ.get("RPT_BENCH_CONFIG5", "7680x4320")
    if (n > 1 or force_dist) and c5 != "0" and args.workload == "bunny" and not animate:
        W5, H5 = (int(x) for x in c5.split("x"))
        if (W5, H5) != (W, H):
            barrier()
            for rr in renderers:
                rr.set_scene_params(scene, W5, H5)
            model5, info5 = rdist.calibrate_split(renderers, scene, W5, H5, rank, n, frames_per_exchange=args.frames_per_exchange, comm=comm)
            cands5 = [None, 0] + sorted({model5 or 4})
            best5, tried5 = rdist.autotune_split(renderers, scene, W5, H5, rank, n, cands5, frames_per_exchange=args.frames_per_exchange,
                                                 force_gather=force_dist, rounds=6, comm=comm)
            name5 = lambda c: "equal" if c is None else ("solo" if c == 0 else f"weighted_root_run_{c}")      # noqa: E731
            config5 = {"workload": f"Scenes/bunny.txt {W5}x{H5} (BASELINE config 5), {n} ranks",
                       "ms_per_frame": {name5(c): round(v * 1e3, 4) for c, v in tried5.items()}, "chosen": name5(best5),
                       "value_chosen": round(W5 * H5 / tried5[best5] / 1e6, 2), "unit": "Mrays/s", "split_calibration": info5,
                       "note": "max over ranks of the wall time per frame, frames in flight; equal = tile k -> rank k mod N, solo = rank 0 renders everything, "
                               "weighted = rank 0 renders its larger share in place"}
            for rr in renderers:
                rr.set_scene_params(scene, W, H)
            barrier()

    if n > 1:
        tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed, kernel_ms_max = float(tt[0]), float(tt[1])
    else:
        kernel_ms_max = kernel_ms

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = W * H / (ms_per_step * 1e-3) / 1e6
        # roofline of the dominant kernel (the render kernel): algorithmic bytes one launch moves on this
        # rank / its mean duration.  With N ranks one launch covers 1/N of the pixels (4 B/px plane).
        if n == 1 and not force_dist:
            alg = algorithmic_bytes(W, H, n_objects)
        elif frame.weighted or frame.solo:
            alg = 16 * W * min(frame.local_rows, H) + 320 * n_objects      # rank 0 writes its rows as 16 B/px framebuffer pixels
        else:
            alg = (16 if frame.full16 else 4) * W * frame.local_rows + 320 * n_objects   # what the render kernel writes (the wire carries plane_bytes/4 of it)
        # Launches of consecutive frames overlap on the device: `overlap` = sum of launch durations / wall time of
        # the region = average number of launches running at once.  A launch's share of the device is then
        # duration / overlap, and achieved = bytes per launch / that (= bytes of all launches / wall time).
        overlap = max(1.0, kernel_sum_ms / (elapsed * 1e3)) if frame.depth > 1 else 1.0
        has_mesh = bool((np.asarray(scene.objects()["type"]) == 2).any())
        achieved = alg / (kernel_ms / overlap * 1e-3) / 1e9
        out = {
            "metric": "Mrays/s (primary rays) on Scenes/bunny.txt at 3840x2160" if (args.workload, W, H) == ("bunny", 3840, 2160)
                      else f"Mrays/s (primary rays) on Scenes/{scene_name}.txt at {W}x{H}",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            # the same region cut into fifths by the COMPLETION times of its launches (HIP events, one clock for all frame slots): the
            # median fifth is what a longer run converges to — a 20-step region is 2 ms long and starts on an idle device
            "ms_per_step_median_of_batches": (lambda v: round(sorted(v)[len(v) // 2], 4) if v else None)(headline_batches),
            "ms_per_step_batches": [round(x, 4) for x in headline_batches],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Scenes/{scene_name}.txt {W}x{H}, camera v={list(vel)} t={t}, interval={scene.params['interval']}, "
                                   f"mesh=Models/bunny.obj (StanfordBunny.obj is missing from the reference)" if scene_name == "bunny"
                       else f"Scenes/{scene_name}.txt {W}x{H}, camera v={list(vel)} t={t}",
                       "frame": "rpt_set_objects + render kernel" + (f" + RCCL gather({frame.plane_bytes} B/px planes, {frame.group} frames per gather) + root scatter" if n > 1 else ""),
                       "frames_in_flight": frame.depth,
                       "sharding": ("none" if n == 1 else "interleaved 8-row tiles, tile k -> rank k mod N" if root_run is None else
                                    "rank 0 renders the whole frame (the exchange would cost more than it saves)" if root_run == 0 else
                                    f"weighted: per {root_run + n - 1} tiles rank 0 renders {root_run} in place, ranks 1..{n - 1} one each into 3 B/px planes"),
                       "split_calibration": split_info,
                       "variant": args.variant},
            # --- the two regimes, side by side, so that they cannot be confused -------------------------------------
            # value / ms_per_step above: THROUGHPUT with `frames_in_flight` frames overlapping on the device (every frame
            # refreshed and rendered completely); ms_per_step is the interval between finished frames, not a latency.
            # blocking: the same frames submitted and waited for one by one, like the reference's runKernel(): LATENCY.
            "regime": f"{frame.depth} frames in flight" if frame.depth > 1 else "one frame at a time",
            "ms_per_frame_blocking": None if blocking_ms is None else round(blocking_ms, 4),
            "value_blocking": None if blocking_ms is None else round(W * H / blocking_ms / 1e3, 2),
            "frame_ms_blocking": spread(blocking_frames),          # host-timed submit+wait of each frame
            "launch_ms_in_flight": spread(launches),               # HIP events around each launch, launches overlapping
            "launch_ms_blocking": spread(blocking_launches),       # the same, one launch at a time
            "kernel_ms": round(kernel_ms, 4),
            "animated": animated,
            "one_frame_at_a_time": None if blocking_ms is None else {
                "ms_per_frame": round(blocking_ms, 4), "value": round(W * H / blocking_ms / 1e3, 2), "kernel_ms": round(blocking_kernel_ms, 4),
                "note": "submit, wait, submit ... like the reference's blocking runKernel(): the frame latency"},
            # The dominant kernel's OWN fraction: algorithmic bytes of one launch / the mean HIP-event duration of that launch when
            # it runs alone on the device (the blocking regime below: what rocprofv3 reports per dispatch for the blocking
            # launches).  The device-level figure with `frames_in_flight` launches overlapping is reported next to it under a name
            # that says what it is; it is a throughput, not a kernel duration.
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": kernel_name, "algorithmic_bytes_per_launch": alg,
                         "launch_ms": round(kernel_ms, 4), "launches_overlapped": round(overlap, 3),
                         "regime": f"{frame.depth} frames in flight: achieved = bytes of all launches / wall time of the region",
                         "definition": "kernel alone: achieved = algorithmic_bytes_per_launch / launch_ms of a launch that has the device "
                                       "to itself (HIP events on its stream = rocprofv3's per-dispatch duration).  device, N in flight: "
                                       "algorithmic_bytes_per_launch / (launch_ms / launches_overlapped) = bytes of all launches / wall time",
                         "note": "16 B/pixel written + 320 B/object read per launch (SURVEY.md §8d); the path is "
                                 "latency/VALU-bound by construction, HBM fraction reported because it is the contract"},
        }
        device_key = f"frac_device_{frame.depth}_in_flight"
        rf = out["roofline"]
        rf[device_key] = rf["frac"]
        rf["device_in_flight"] = {"achieved": rf["achieved"], "frac": rf["frac"], "launch_ms": rf["launch_ms"], "launches_overlapped": rf["launches_overlapped"],
                                  "kernel": kernel_name, "note": rf["regime"], "traffic": None}
        short = lambda label: label.split(" ")[0]          # noqa: E731  ("rpt_render_kernel_ballot_w5 (41; ...)" -> the kernel's name)
        rf["describes"] = f"device level, {frame.depth} frames in flight ({short(kernel_name)})"
        if blocking_kernel_ms:
            a1 = alg / (blocking_kernel_ms * 1e-3) / 1e9
            # roofline.frac / achieved / launch_ms / kernel / traffic are the KERNEL-ALONE figures (the contract's "dominant kernel"
            # fraction): ONE kernel in ONE regime.  `value` and ms_per_step come from the frames in flight; their device-level
            # fraction, kernel and traffic are the block roofline.device_in_flight.  (r02 -> r03 changed what the top-level keys
            # mean; `describes` says it in the line itself.)
            rf.update({"achieved": round(a1, 2), "frac": round(a1 / HBM_PEAK_GBS, 5), "frac_kernel_alone": round(a1 / HBM_PEAK_GBS, 5),
                       "launch_ms": round(blocking_kernel_ms, 4), "launches_overlapped": 1.0,
                       "kernel": blocking_kernel_name,
                       "describes": f"the kernel alone: {short(blocking_kernel_name)}, one launch at a time (the blocking rpt_render); value / ms_per_step are "
                                    f"the {frame.depth} frames in flight, see roofline.device_in_flight ({short(kernel_name)})",
                       "regime": "one launch at a time, nothing overlapped (the blocking rpt_render): algorithmic bytes / the launch's own HIP-event duration"})
            rf["frac_blocking"] = rf["frac"]          # (round 2's name for the same number)
        else:
            rf["frac_kernel_alone"] = None
            rf["note_regime"] = "N > 1: no kernel-alone measurement in this run; frac is the device-level figure"
        # HBM traffic: rocprofv3 PMC passes of this command, quoted only when taken on THIS build, per kernel — the block of the
        # kernel each figure names
        if n == 1 and not force_dist and args.variant == 0:
            traffic, source = measured_traffic(args.workload, W, H, short(rf["kernel"]))
            rf["traffic"] = traffic
            rf["traffic_source"] = (f"{source}, block {short(rf['kernel'])} (rocprofv3 --pmc of this command on this build of librpt_hip.so: recorded hash matches)"
                                    if source else "no committed PMC summary was taken on this build of librpt_hip.so")
            rf["device_in_flight"]["traffic"], _ = measured_traffic(args.workload, W, H, short(kernel_name))
            # beside the contract's HBM figure: how much of the vector ALUs' issue rate the same launches use (the kernel alone over its
            # own duration; the frames in flight over the interval between finished frames) — null without counters of this build
            rf["valu"] = measured_valu(args.workload, W, H, short(rf["kernel"]), rf["launch_ms"] * 1e-3)
            rf["device_in_flight"]["valu"] = measured_valu(args.workload, W, H, short(kernel_name), ms_per_step * 1e-3)
        if n > 1 or force_dist:
            out["comm"] = {"backend": td.get_backend(), "world_size": n, "ranks_in_group": td.get_world_size(),
                           "exchange": "ncclGather through ctypes (relativitypathtracer_amd/rccl.py), one per " + (f"{frame.group} frames" if frame.group > 1 else "frame") if comm is not None
                                       else "torch.distributed.gather",
                           "split_timings_ms_per_frame": split_timings,
                           "config5": config5,
                           "note": "every N > 1 figure before an 8-GPU node has run this line is a rehearsal on one GPU or a model: none is a measurement of xGMI"}
        if not args.no_cpu_baseline and n == 1:
            out["cpu_baseline"] = cpu_baseline(scene, W, H)
            # second column of SURVEY.md §8(d): primary + shadow rays, the shadow rays counted by the oracle
            shadow = out["cpu_baseline"]["shadow_rays_per_frame"]
            out["total_rays"] = {"primary_per_frame": W * H, "shadow_per_frame": shadow,
                                 "value": round((W * H + shadow) / (ms_per_step * 1e-3) / 1e6, 2), "unit": "Mrays/s"}
        if args.check:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_ffi
            # every framebuffer that received frames: the root's, or (no exchange) the one of each slot in flight
            fbs = [frame.framebuffer] if (frame.exchange or animate) else [sl.framebuffer for sl in frame.slots if sl.frames]   # animated: the slots hold different instants, only the last frame is the scene's current state
            fbs = [f.cpu().numpy().view(np.uint8).reshape(H, W, 16) for f in fbs]
            worst, differing = 0, 0
            for (r0, r1) in [(0, 8), (H * 2 // 5, H * 2 // 5 + 16), (H // 2, H // 2 + 16), (H - 8, H)]:
                opx, _, _ = oracle_ffi.render(scene, W, H, rows=(r0, r1), want_rgb=False)
                want = opx["rgba"].reshape(H, W, 4)[r0:r1].astype(np.int16)
                for fb in fbs:
                    d = np.abs(fb[r0:r1, :, 8:12].astype(np.int16) - want)
                    worst, differing = max(worst, int(d.max())), differing + int((d > 0).sum())
            # every scene is expected to be identical; a 1-LSB difference is reported as such rather than as a mismatch
            out["check"] = ("framebuffer rows identical to the oracle" if worst == 0 else
                            f"framebuffer rows within 1 LSB of the oracle ({differing} bytes differ)" if worst == 1 else "MISMATCH vs oracle")
        print(json.dumps(out), flush=True)
    if n > 1 or force_dist:
        td.barrier()
        if comm is not None:
            torch.cuda.synchronize()
            comm.destroy()
        td.destroy_process_group()
    for rr in renderers:
        rr.close()


if __name__ == "__main__":
    main()
