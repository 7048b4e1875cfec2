#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/profile.sh into one JSON (the file bench.py reads for
roofline.traffic):  python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<name>_pmc_summary.json

ONE BLOCK PER KERNEL, keyed by the kernel's full name (round 3 bucketed every kernel whose name contains "rpt_render_kernel"
under "render": a bench.py run launches the throughput kernel with frames in flight AND the latency kernel one launch at a
time, and their counters ended up averaged together — VERDICT r03).  Per kernel: which regime it ran in during a bench.py run,
mean/min/max of every counter over its launches, register counts, and the derived HBM bytes per launch — FETCH_SIZE and
WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE counts 64-B requests as 32 B, so reads are doubled (MI355X_MICROARCH.md, HBM /
rocprofv3 section).
"""
import collections
import csv
import glob
import json
import os
import sys

# how bench.py uses the render kernels (bench.py: timed region = rpt_render_async on `frames_in_flight` contexts; then the blocking
# rpt_render on one): frames of more than RPT_LATENCY_KERNEL_MAX_PIXELS get kernel 41 in flight and 43 blocking, smaller ones 43 in both
REGIMES = {
    "rpt_render_kernel_ballot_w5": "kernel 41: frames in flight (launches overlap on the device)",
    "rpt_render_kernel_ballot_first_w5": "kernel 43: the blocking rpt_render, one launch at a time (and frames of at most 3 Mpx in flight as well: both regimes are in these launches)",
    "rpt_render_kernel_analytic_w8": "kernel 44: scenes without meshes, both regimes",
    "rpt_render_kernel_unculled_w5": "kernel 3: un-culled (rpt_verify_frame, frames wider than 4 : 1)",
}


def short_name(kernel_name):
    """'rptd::rpt_render_kernel_ballot_w5(rptd::KernelArgs)' -> 'rpt_render_kernel_ballot_w5'"""
    n = kernel_name.split("(")[0].strip()
    return n.split("::")[-1].split(" ")[-1]


def build_id():
    """What the counters were taken on: the git revision (when the snapshot has one) and the hash of librpt_hip.so —
    bench.py quotes roofline.traffic only from a summary whose library hash equals the running library's."""
    import hashlib
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "relativitypathtracer_amd", "librpt_hip.so")
    ident = {"librpt_hip_sha256": hashlib.sha256(open(so, "rb").read()).hexdigest() if os.path.exists(so) else None, "git_rev": None}
    try:
        ident["git_rev"] = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        pass
    return ident


def main(src, dst, command=None):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = short_name(row["Kernel_Name"])
                if not key.startswith("rpt_"):
                    continue
                per[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta.setdefault(key, {"kernel": row["Kernel_Name"], "regime_in_a_bench_run": REGIMES.get(key, "helper kernel"),
                                      "VGPR_Count": row["VGPR_Count"], "Accum_VGPR_Count": row["Accum_VGPR_Count"], "SGPR_Count": row["SGPR_Count"],
                                      "Grid_Size": row["Grid_Size"], "Workgroup_Size": row["Workgroup_Size"], "Scratch_Size": row["Scratch_Size"]})
    kernels = {}
    for key in sorted(meta):
        blk = dict(meta[key])
        for c, v in sorted(per[key].items()):
            blk[c] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}

        def mean(c, blk=blk):
            return blk.get(c, {}).get("mean", 0.0)
        hit, miss = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
        blk["derived"] = {
            "hbm_read_bytes_per_launch (FETCH_SIZE KB x1024 x2 gfx950 correction)": mean("FETCH_SIZE") * 1024 * 2,
            "hbm_write_bytes_per_launch (WRITE_SIZE KB x1024)": mean("WRITE_SIZE") * 1024,
            "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
            # (rounds 2-4 divided by SQ_ACTIVE_INST_VALU and by 4 here and recorded a quarter of the figure — the tables in DESIGN.md
            # always used THREAD_CYCLES / INSTS, "of 64"; the committed r04 summaries were corrected in place by tools/fix_r04_lanes.py)
            "lanes_active_per_valu_instruction": (mean("SQ_THREAD_CYCLES_VALU") / mean("SQ_INSTS_VALU")) if mean("SQ_INSTS_VALU") else None,
            "valu_wave_instructions_per_launch": mean("SQ_INSTS_VALU") or None,
            "wave_life_waiting_share": (mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES")) if mean("SQ_WAVE_CYCLES") else None,
        }
        kernels[key] = blk
    out = {"kernels": kernels, "build": build_id(), "command": command,
           "note": "one block per kernel (full name); bench.py's roofline.traffic is the block of the kernel roofline.kernel names"}
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v["derived"] for k, v in kernels.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
