#!/usr/bin/env python3
"""After `gpurun … bash tools/profile.sh r02_<workload>_<W>x<H> …` has left its output under gpurun_out/prof_<tag>/: copy the two
condensed files of every tag into profiles/ and print one table row per workload (DESIGN.md §6.2's counters table).

    python tools/collect_profiles.py r02_bunny_3840x2160 r02_shadows_3840x2160 …
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tags):
    for t in tags:
        src = os.path.join(ROOT, "gpurun_out", f"prof_{t}")
        for suf in ("_pmc_summary.json", "_kernel_stats.csv"):
            shutil.copy(os.path.join(src, t + suf), os.path.join(ROOT, "profiles", t + suf))
        d = json.load(open(os.path.join(ROOT, "profiles", t + "_pmc_summary.json")))
        for name, r in d["kernels"].items():          # one block per kernel (tools/pmc_summary.py)
            if not name.startswith("rpt_render_kernel"):
                continue

            def g(c, r=r):
                return r.get(c, {}).get("mean", 0.0)
            wc = max(g("SQ_WAVE_CYCLES"), 1.0)
            print(f"{t}  {name}  lib {d['build']['librpt_hip_sha256'][:10]}  | {g('SQ_INSTS_VALU') / 1e6:.1f} M | "
                  f"{g('SQ_THREAD_CYCLES_VALU') / max(g('SQ_INSTS_VALU'), 1):.0f} of 64 | "
                  f"{100 * g('SQ_WAIT_ANY') / wc:.0f} % / {100 * g('SQ_WAIT_INST_ANY') / wc:.0f} % / {100 * g('SQ_ACTIVE_INST_ANY') / wc:.0f} % | "
                  f"{100 * (1 - g('TCP_TCC_READ_REQ_sum') / max(g('TCP_TOTAL_CACHE_ACCESSES_sum'), 1)):.1f} % | {r['derived']['l2_hit_rate'] or 0:.2f} | "
                  f"{g('WRITE_SIZE') * 1024 / 1e6:.1f} / {g('FETCH_SIZE') * 2048 / 1e6:.1f} |   launches {int(r.get('SQ_WAVES', {}).get('launches', 0))}")
        for line in open(os.path.join(ROOT, "profiles", t + "_kernel_stats.csv")).read().splitlines()[1:3]:
            print("     ", line[:130])


if __name__ == "__main__":
    main(sys.argv[1:])
