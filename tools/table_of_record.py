#!/usr/bin/env python3
"""DESIGN.md section 6.1's rows from profiles/<round>_bench_*.json, <round>_*_kernel_stats.csv and <round>_configs_default.txt:
the table of record is generated, not typed.   usage: python tools/table_of_record.py [r03]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROWS = [("**bunny 3840×2160 (the metric)**", "bunny_3840x2160"), ("bunny 1920×1080 (config 3)", "bunny_1920x1080"),
        ("shadows 3840×2160, t = 16 (config 4)", "shadows_3840x2160"), ("arch 1920×1080, v = 0.95c (config 2)", "arch_1920x1080"),
        ("cube 640×480 (config 1)", "cube_640x480"), ("bunny 7680×4320 (config 5, one GPU)", "bunny_7680x4320"),
        ("cubes 3840×2160 (34 objects, camera 0.3c)", "cubes_3840x2160")]


def rocprof_blocking_ms(key):
    """average of the blocking kernel where the trace holds it apart from the launches in flight (4K and 8K mesh frames)"""
    path = os.path.join(ROOT, "profiles", f"{rnd}_{key}_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    names = {}
    for row in csv.reader(open(path)):
        if len(row) > 3 and "rpt_render_kernel" in row[0]:
            names[row[0]] = float(row[3]) / 1e6
    first = [v for k, v in names.items() if "ballot_first" in k]
    return first[0] if first and len(names) > 1 else None


def sp(x):
    return f"{x:,.0f}".replace(",", " ")


for label, key in ROWS:
    d = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_{key}.json")))
    r = d["roofline"]
    rp = rocprof_blocking_ms(key)
    alone = f"{r['launch_ms']:.4f}" + (f" ({rp:.4f})" if rp else "")
    traffic = f"{r['traffic'] / 1e6:.1f}" if r.get("traffic") else "—"
    print(f"| {label} | {sp(d['value'])} | {d['ms_per_step']:.4f} | {d['ms_per_frame_blocking']:.4f} | {alone} | {r['frac']:.4f} | "
          f"{r['device_in_flight']['frac']:.3f} | {r['algorithmic_bytes_per_launch'] / 1e6:.1f} | {traffic} | {d['animated']['ms_per_step']:.4f} | "
          f"{d['cpu_baseline']['value']:.1f} |")
have = {k for _, k in ROWS}
for line in open(os.path.join(ROOT, "profiles", f"{rnd}_configs_default.txt")):
    f = line.split()
    if len(f) < 14 or f"{f[0]}_{f[1]}" in have:
        continue
    depth = f[10]
    print(f"| {f[0]} {f[1].replace('x', '×')} ({depth} slots, `configs.py`) | {sp(float(f[-2]))} | {float(f[13]):.4f} | {float(f[8]):.4f} | | | | "
          f"{16 * int(f[1].split('x')[0]) * int(f[1].split('x')[1]) / 1e6:.1f} | | | |")
