#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/profile.sh into one JSON (the file bench.py reads for
roofline.traffic):  python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<name>_pmc_summary.json

Per kernel of the render path: mean/min/max of every counter over the launches, register counts, and the derived
HBM bytes per launch — FETCH_SIZE and WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE counts 64-B requests as
32 B, so reads are doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = {"tile_bin": "rpt_tile_bin_kernel", "render": "rpt_render_kernel", "shade": "rpt_shade_kernel"}


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
    per = {k: collections.defaultdict(list) for k in KERNELS}
    meta = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                for key, pat in KERNELS.items():
                    if pat in row["Kernel_Name"]:
                        per[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                        meta.setdefault(key, {"kernel": row["Kernel_Name"], "VGPR_Count": row["VGPR_Count"],
                                              "Accum_VGPR_Count": row["Accum_VGPR_Count"], "SGPR_Count": row["SGPR_Count"],
                                              "Grid_Size": row["Grid_Size"], "Workgroup_Size": row["Workgroup_Size"],
                                              "Scratch_Size": row["Scratch_Size"]})
    out = {}
    for key in KERNELS:
        if key not in meta:
            continue
        out[key] = dict(meta[key])
        for c, v in sorted(per[key].items()):
            out[key][c] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}

    def mean(key, c):
        return out.get(key, {}).get(c, {}).get("mean", 0.0)
    rd = sum(mean(k, "FETCH_SIZE") for k in KERNELS) * 1024 * 2
    wr = sum(mean(k, "WRITE_SIZE") for k in KERNELS) * 1024
    hit, miss = mean("render", "TCC_HIT_sum"), mean("render", "TCC_MISS_sum")
    out["build"] = build_id()
    out["command"] = command
    out["derived"] = {
        "hbm_read_bytes_per_launch (FETCH_SIZE KB x1024 x2 gfx950 correction)": rd,
        "hbm_write_bytes_per_launch (WRITE_SIZE KB x1024)": wr,
        "l2_hit_rate_render": hit / (hit + miss) if hit + miss else None,
    }
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["derived"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
