#!/usr/bin/env python3
"""Fraction of the vector ALUs' issue rate each line of the record uses: SQ_INSTS_VALU per launch (profiles/<round>_<workload>_
<W>x<H>_pmc_summary.json, the block of the kernel) over 1 024 SIMD-32s x 2.4 GHz / 2 cycles per wave64 instruction x the launch's
time from the bench line of the same library (profiles/<round>_bench_<workload>_<W>x<H>.json): the kernel alone over its own
duration, the frames in flight over the interval between finished frames.  The same arithmetic as bench.py's roofline.valu.

    python tools/valu_fraction.py r04          # one markdown row per workload
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import VALU_PEAK_WAVE_INSTR_PER_S as PEAK  # noqa: E402


def main(rnd):
    print("| workload | kernel alone | wave-instr / launch | launch ms | VALU issue | lanes of 64 | in flight | wave-instr / launch | ms / step | VALU issue | lanes of 64 |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}_bench_*_*x*.json"))):
        m = re.match(rf"{rnd}_bench_(\w+?)_(\d+)x(\d+)\.json", os.path.basename(path))
        if not m:
            continue
        line = json.loads(open(path).read().strip().splitlines()[-1])
        summ = os.path.join(ROOT, "profiles", f"{rnd}_{m.group(1)}_{m.group(2)}x{m.group(3)}_pmc_summary.json")
        if not os.path.exists(summ) or "roofline" not in line:
            continue
        pmc = json.load(open(summ))
        rf = line["roofline"]
        cells = [f"{m.group(1)} {m.group(2)}×{m.group(3)}"]
        for label, seconds in ((rf["kernel"], rf["launch_ms"] * 1e-3), (rf["device_in_flight"]["kernel"], line["ms_per_step"] * 1e-3)):
            k = label.split(" ")[0]
            blk = pmc["kernels"].get(k)
            if blk is None:
                cells += [k.replace("rpt_render_kernel_", ""), "—", f"{seconds * 1e3:.4f}", "—", "—"]
                continue
            n, t = blk["SQ_INSTS_VALU"]["mean"], blk["SQ_THREAD_CYCLES_VALU"]["mean"]
            cells += [k.replace("rpt_render_kernel_", ""), f"{n / 1e6:.1f} M", f"{seconds * 1e3:.4f}", f"{100 * n / (PEAK * seconds):.0f} %", f"{min(t / n, 64.0):.0f}"]
        print("| " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
