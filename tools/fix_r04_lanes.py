#!/usr/bin/env python3
"""One-off: the derived field lanes_active_per_valu_instruction of the committed round-4 PMC summaries was a quarter of the real
figure (tools/pmc_summary.py divided SQ_THREAD_CYCLES_VALU by SQ_ACTIVE_INST_VALU and by 4).  Recomputes it from the counter
means the summaries hold themselves (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU: lanes of 64) and adds
valu_wave_instructions_per_launch, which bench.py's roofline.valu reads.  Counters, build hash and command are left as recorded."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_*_pmc_summary.json"))):
    d = json.load(open(path))
    for name, blk in d.get("kernels", {}).items():
        n = blk.get("SQ_INSTS_VALU", {}).get("mean")
        t = blk.get("SQ_THREAD_CYCLES_VALU", {}).get("mean")
        blk["derived"]["lanes_active_per_valu_instruction"] = (t / n) if n and t else None
        blk["derived"]["valu_wave_instructions_per_launch"] = n or None
    json.dump(d, open(path, "w"), indent=1)
    print(os.path.basename(path), {k: round(v["derived"]["lanes_active_per_valu_instruction"] or 0, 1) for k, v in d["kernels"].items() if k.startswith("rpt_render")})
