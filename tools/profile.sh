#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3; run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash tools/profile.sh r01'
# Writes raw output under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
# counters: separate passes, --kernel-trace only (never combined with sys/hip/hsa tracing)
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  NAME=$(echo "$SET" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- $BENCH > "$OUT/pmc_$NAME.log" 2>&1 || { echo "pmc pass $SET failed"; tail -5 "$OUT/pmc_$NAME.log"; }
done
find "$OUT" -name "*.csv" | head -50
