#!/usr/bin/env python3
"""three quotients by one scalar through ONE refined reciprocal (csrc/rpt_device_math.hip.h).
Part 1 — bit-identity, on the device: rpt_probe_division, ~10^9 (x, y, z, s) sets per mode.
Part 2 — what it buys: bench.py lines A/B/A/B, the product library against the experiment builds librpt_hip_rcp1.so / _rcp2.so
(RPT_HIP_LIB), each with --check (rows of the last frame against the oracle).
usage (GPU box): python tools/r04_rcp_ab.py > gpurun_out/r04/rcp_ab.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def probe():
    from relativitypathtracer_amd.renderer import Renderer
    r = Renderer(0)
    names = {0: "random significands, exponents across and beyond the domain", 1: "denominators with an all-ones significand",
             2: "normalize(): s = sqrt(dot(v, v))", 3: "arbitrary bit patterns (guarded form only)"}
    ok = True
    for mode in (0, 1, 2, 3):
        tot = [0, 0, 0, 0, 0]
        first = None
        for seed in range(4):
            counts, samples = r.probe_division(mode, 1234 + 977 * seed, 4096, 256, 8)       # 4096 * 256 * 256 = 2.7e8 sets per call
            tot = [a + b for a, b in zip(tot, counts)]
            if counts[4] and first is None:
                first = samples[: min(4, counts[4])].tolist()
        sets = 4 * 4096 * 256 * 256
        print(f"mode {mode} ({names[mode]}): {sets:.3g} sets, {tot[0]:.4g} inside the fast path's domain = {3 * tot[0]:.4g} quotients; "
              f"mismatches with ONE residual correction {tot[1]}, with TWO {tot[2]}; guarded form over all sets {tot[3]}", flush=True)
        if first:
            print("   first mismatching sets (x, y, z, s):", first, flush=True)
        ok = ok and tot[2] == 0 and tot[3] == 0
    r.close()
    return ok


def bench(lib, workload, w, h, steps=60):
    env = dict(os.environ)
    if lib:
        env["RPT_HIP_LIB"] = os.path.join(ROOT, "relativitypathtracer_amd", lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "10", "--workload", workload, "--width", str(w), "--height", str(h),
                        "--no-cpu-baseline", "--check"], capture_output=True, text=True, env=env, timeout=600)
    if p.returncode != 0:
        return None, p.stderr[-400:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    return d, None


def main():
    ok = probe()
    print("bit-identical on everything compared (two corrections, and the guarded form)" if ok else "NOT bit-identical: see above", flush=True)
    libs = [(None, "product (3 IEEE divisions)"), ("librpt_hip_rcp1.so", "shared reciprocal, 1 correction"), ("librpt_hip_rcp2.so", "shared reciprocal, 2 corrections")]
    for workload, w, h in (("shadows", 3840, 2160), ("arch", 1920, 1080), ("cubes", 3840, 2160), ("bunny", 3840, 2160), ("bunny", 1920, 1080)):
        for rep in range(2):
            for lib, label in libs:
                if lib and not os.path.exists(os.path.join(ROOT, "relativitypathtracer_amd", lib)):
                    continue
                d, err = bench(lib, workload, w, h)
                if d is None:
                    print(f"{workload} {w}x{h} {label}: FAILED {err}", flush=True)
                    continue
                print(f"{workload:8s} {w}x{h} run {rep} {label:36s}: {d['ms_per_step']:.4f} ms/frame in flight, {d.get('ms_per_frame_blocking')} one at a time, check: {d.get('check')}", flush=True)


if __name__ == "__main__":
    main()
