#!/bin/bash
# round 4 soak on the certified bounds and the derived shadow culls: rpt_verify_frame, kernels 41, 43 and the default selection
# against the un-culled kernel — generated scenes of every kind, animated sweeps of every shipped scene
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/verify_soak.txt
: > $OUT
FIRST=${1:-200000}; LAST=${2:-204000}
for kind in ${KINDS:-random extreme close walls ellipsoids meshwalls}; do
  timeout -k 10 ${KIND_TIMEOUT:-400} python tools/verify_fuzz.py --first $FIRST --last $LAST --kinds $kind >> $OUT 2>&1 || { tail -5 $OUT; exit 1; }
  tail -2 $OUT
done
for p in ${SWEEPS:-21 22 23}; do
  echo "== sweep path $p, 4000 states, 1280x720" >> $OUT
  timeout -k 10 300 python tools/verify_sweep.py --states 4000 --width 1280 --height 720 --path $p >> $OUT 2>&1 || { tail -5 $OUT; exit 1; }
  tail -1 $OUT
done
