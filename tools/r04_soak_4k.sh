#!/bin/bash
# the same verifications at 3840x2160, where the default selection IS kernel 41 (rpt_render_async's choice above 3 Mpx)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/verify_soak_4k.txt
: > $OUT
for kind in extreme walls ellipsoids; do
  timeout -k 10 300 python tools/verify_fuzz.py --first 600000 --last 620000 --kinds $kind --size 3840x2160 >> $OUT 2>&1 || { tail -5 $OUT; exit 1; }
  tail -2 $OUT
done
for kind in random close meshwalls; do
  timeout -k 10 300 python tools/verify_fuzz.py --first 600000 --last 602000 --kinds $kind --size 3840x2160 >> $OUT 2>&1 || { tail -5 $OUT; exit 1; }
  tail -2 $OUT
done
echo "== sweep path 26, 1500 states, 3840x2160" >> $OUT
timeout -k 10 300 python tools/verify_sweep.py --states 1500 --width 3840 --height 2160 --path 26 >> $OUT 2>&1 || { tail -5 $OUT; exit 1; }
tail -10 $OUT
