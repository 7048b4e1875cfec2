#!/usr/bin/env python3
"""Exit faces of a leaf packed into one register (csrc/rpt_kernels.hip.h PackedExitPlan, `make exitfaces`) against the three hoisted
face ids the compiler spills and reloads inside the leaf loop of kernel 41: bench.py lines A/B/A/B/A/B, the product library against
the experiment build (RPT_HIP_LIB=librpt_hip_exitfaces.so), each with --check (rows of the last frame against the oracle).  Every
line carries the frames in flight and the blocking kernel both.  Result: profiles/r04_exitplan_ab.txt (not adopted).
usage (GPU box): make -C relativitypathtracer_amd/csrc exitfaces; python tools/r04_exitplan_ab.py > gpurun_out/r04/exitplan_ab.txt"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPERIMENT = "librpt_hip_exitfaces.so"


def sha(lib):
    return hashlib.sha256(open(os.path.join(ROOT, "relativitypathtracer_amd", lib), "rb").read()).hexdigest()[:16]


def bench(lib, workload, w, h, inflight, steps):
    env = dict(os.environ)
    if lib:
        env["RPT_HIP_LIB"] = os.path.join(ROOT, "relativitypathtracer_amd", lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "10", "--workload", workload, "--width", str(w), "--height", str(h),
                        "--inflight", str(inflight), "--no-cpu-baseline", "--check"], capture_output=True, text=True, env=env, timeout=600)
    if p.returncode != 0:
        return None, p.stderr[-400:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]), None


def main():
    print(f"before: librpt_hip.so {sha('librpt_hip.so')}   after: {EXPERIMENT} {sha(EXPERIMENT)}", flush=True)
    libs = [(None, "before"), (EXPERIMENT, "after ")]
    for workload, w, h, inflight, steps in (("bunny", 3840, 2160, 4, 200), ("bunny", 1920, 1080, 4, 200), ("bunny", 7680, 4320, 4, 60),
                                            ("shadows", 3840, 2160, 4, 200), ("arch", 1920, 1080, 4, 200), ("cubes", 3840, 2160, 4, 200)):
        med = {}
        for rep in range(3):
            for lib, label in libs:
                d, err = bench(lib, workload, w, h, inflight, steps)
                if d is None:
                    print(f"{workload} {w}x{h} inflight {inflight} {label}: FAILED {err}", flush=True)
                    continue
                m = d.get("ms_per_step_median_of_batches") or d["ms_per_step"]
                alone = (d.get("one_frame_at_a_time") or {}).get("kernel_ms")
                med.setdefault(label, []).append((m, alone or 0.0))
                print(f"{workload} {w}x{h} {label}: in flight ms/step {d['ms_per_step']:.4f} (median of batches {m:.4f})   one at a time, kernel {alone} ms   "
                      f"value {d['value']:.0f}  check: {d.get('check')}", flush=True)
        if len(med) == 2:
            for k, what in ((0, "in flight, ms/step (median of batches)"), (1, "kernel alone, ms")):
                a, b = sorted(x[k] for x in med["before"])[1], sorted(x[k] for x in med["after "])[1]
                print(f"   => {what}, median of three runs: before {a:.4f}  after {b:.4f}  ({100 * (b - a) / max(a, 1e-9):+.2f} %)", flush=True)


if __name__ == "__main__":
    main()
