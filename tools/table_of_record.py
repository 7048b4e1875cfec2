#!/usr/bin/env python3
"""DESIGN.md section 6.1's rows from profiles/<round>_bench_*.json, <round>_*_kernel_stats.csv and <round>_configs_default.txt, and
section 6.2's counter rows from profiles/<round>_*_pmc_summary.json: the tables of record are generated, not typed.
usage: python tools/table_of_record.py [r03] [--write]     (--write: replace the rows between the markers in DESIGN.md and the
library hash in its heading and in profiles/README.md; every file must have been taken on ONE build of librpt_hip.so)"""
import re
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rnd = args[0] if args else "r04"
out_rows, counter_rows, hashes = [], [], set()
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
    return first[0] if first and len([k for k in names if "ballot" in k or "analytic" in k]) > 1 else None


def sp(x):
    return f"{x:,.0f}".replace(",", " ")


for label, key in ROWS:
    d = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_{key}.json")))
    r = d["roofline"]
    rp = rocprof_blocking_ms(key)
    alone = f"{r['launch_ms']:.4f}" + (f" ({rp:.4f})" if rp else "")
    traffic = f"{r['traffic'] / 1e6:.1f}" if r.get("traffic") else "—"
    out_rows.append(f"| {label} | {sp(d['value'])} | {d['ms_per_step']:.4f} | {d['ms_per_frame_blocking']:.4f} | {alone} | {r['frac']:.4f} | "
          f"{r['device_in_flight']['frac']:.3f} | {r['algorithmic_bytes_per_launch'] / 1e6:.1f} | {traffic} | {d['animated']['ms_per_step']:.4f} | "
          f"{d['cpu_baseline']['value']:.1f} |")
have = {k for _, k in ROWS}
for line in open(os.path.join(ROOT, "profiles", f"{rnd}_configs_default.txt")):
    f = line.split()
    if len(f) < 14 or f"{f[0]}_{f[1]}" in have:
        continue
    depth = f[10]
    out_rows.append(f"| {f[0]} {f[1].replace('x', '×')} ({depth} slots, `configs.py`) | {sp(float(f[-2]))} | {float(f[13]):.4f} | {float(f[8]):.4f} | | | | "
          f"{16 * int(f[1].split('x')[0]) * int(f[1].split('x')[1]) / 1e6:.1f} | | | |")

COUNTERS = [("bunny 4K", "bunny_3840x2160", ""), ("bunny 1080p", "bunny_1920x1080", ""), ("shadows 4K", "shadows_3840x2160", ""),
            ("arch 1080p", "arch_1920x1080", ""), ("cube 640×480", "cube_640x480", ""), ("bunny 8K", "bunny_7680x4320", ""),
            ("cubes 4K (34 textured cubes)", "cubes_3840x2160", " (the fetch is the texture pool)")]
KERNEL_ROWS = [("rpt_render_kernel_ballot_w5", "kernel 41, frames in flight"), ("rpt_render_kernel_ballot_first_w5", "kernel 43"),
               ("rpt_render_kernel_analytic_w8", "kernel 44, both regimes")]
for label, key, note in COUNTERS:
    path = os.path.join(ROOT, "profiles", f"{rnd}_{key}_pmc_summary.json")
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    hashes.add(d["build"]["librpt_hip_sha256"])
    alg = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_{key}.json")))["roofline"]["algorithmic_bytes_per_launch"] / 1e6
    blocks = d["kernels"]
    for kname, regime in KERNEL_ROWS:
        if kname not in blocks:
            continue
        r = blocks[kname]
        if kname.endswith("first_w5"):
            regime = "kernel 43, one at a time" if "rpt_render_kernel_ballot_w5" in blocks else "kernel 43, in flight AND one at a time (frame <= 3 Mpx)"

        def g(c, r=r):
            return r.get(c, {}).get("mean", 0.0)
        wc = max(g("SQ_WAVE_CYCLES"), 1.0)
        lanes = g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_INSTS_VALU"), 1)
        lanes_s = f"{min(lanes, 64):.0f} of 64" + (f" (the ratio reads {lanes:.0f}: instructions that issue over eight cycles count twice)" if lanes > 64.5 else "")
        counter_rows.append(f"| {label} — {regime} ({int(r.get('SQ_WAVES', {}).get('launches', 0))} launches) | {g('SQ_INSTS_VALU') / 1e6:.1f} M | {lanes_s} | "
                            f"{100 * g('SQ_WAIT_ANY') / wc:.0f} % / {100 * g('SQ_WAIT_INST_ANY') / wc:.0f} % / {100 * g('SQ_ACTIVE_INST_ANY') / wc:.0f} % | "
                            f"{100 * (1 - g('TCP_TCC_READ_REQ_sum') / max(g('TCP_TOTAL_CACHE_ACCESSES_sum'), 1)):.1f} % | {r['derived']['l2_hit_rate'] or 0:.2f} | "
                            f"{g('WRITE_SIZE') * 1024 / 1e6:.1f} / {g('FETCH_SIZE') * 2048 / 1e6:.1f} vs {alg:.1f}{note} |")
print("\n".join(out_rows))
print()
print("\n".join(counter_rows))
print("library:", ", ".join(sorted(h[:16] for h in hashes)))
if "--write" in sys.argv:
    if len(hashes) != 1:
        sys.exit("the PMC summaries were taken on more than one build of librpt_hip.so: not writing")
    h16 = next(iter(hashes))[:16]
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    # the markers stand before the table's header and after its last row (a comment between rows would end the table)
    text = re.sub(r"(<!-- table-of-record rows:[^\n]*-->\n\|[^\n]*\n\|---[^\n]*\n).*?(<!-- /table-of-record rows -->)", lambda m: m.group(1) + "\n".join(out_rows) + "\n" + m.group(2), text, flags=re.S)
    text = re.sub(r"(<!-- counter rows:[^\n]*-->\n\|[^\n]*\n\|---[^\n]*\n).*?(<!-- /counter rows -->)", lambda m: m.group(1) + "\n".join(counter_rows) + "\n" + m.group(2), text, flags=re.S)
    text = re.sub(r"(### 6\.1 Table of record — library `)[0-9a-f]+(…`)", lambda m: m.group(1) + h16 + m.group(2), text)
    open(path, "w").write(text)
    path = os.path.join(ROOT, "profiles", "README.md")
    whole = open(path).read()
    cut = whole.rindex("## Round ")            # only the current round's section is rewritten
    keep, text = whole[:cut], whole[cut:]
    text = re.sub(r"(Library of record: `librpt_hip.so` sha256 `)[0-9a-f]+(…`)", lambda m: m.group(1) + h16 + m.group(2), text)
    d = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_bunny_3840x2160.json")))
    r = d["roofline"]
    line = (f"(bunny 4K, {d['config'].get('frames_in_flight', 4)} frames in flight: {sp(d['value'])} Mrays/s, {d['ms_per_step']:.4f} ms/step; "
            f"{d['ms_per_frame_blocking']:.4f} ms one at a time; kernel alone {r['launch_ms']:.4f} ms = {r['frac']:.4f} of the roofline; HBM traffic "
            f"{r['traffic'] / 1e6:.1f} MB vs {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic; animated {d['animated']['ms_per_step']:.4f} ms/step; "
            f"CPU oracle {d['cpu_baseline']['value']:.1f} Mrays/s on {d['cpu_baseline']['cores']} threads)")
    text = re.sub(r"\(bunny 4K, \w+ frames in flight: [^)]*threads\)", lambda m: line, text)
    rp = rocprof_blocking_ms("bunny_3840x2160")
    if rp:
        text = re.sub(r"(the cold first one included\) )[0-9.]+( µs average)", lambda m: m.group(1) + f"{rp * 1e3:.1f}" + m.group(2), text)
    open(path, "w").write(keep + text)
    print("written: DESIGN.md sections 6.1 / 6.2, profiles/README.md")
