#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3; run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash tools/profile.sh r02_bunny4k [bench.py arguments, e.g. --workload shadows]'
# Writes raw output under gpurun_out/prof_<tag>/ and the two condensed files the judge reads into profiles/:
#   profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats)
#   profiles/<tag>_pmc_summary.json   (tools/pmc_summary.py over the PMC passes; carries the git revision and the
#                                      hash of librpt_hip.so, which bench.py checks before quoting roofline.traffic)
# (gpurun merges gpurun_out/ back; profiles/ on the box does not travel, so both are also left in gpurun_out/prof_<tag>/.)
set -o pipefail
TAG=${1:-r02}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline $*"
echo "$BENCH" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
# counters: separate passes, --kernel-trace only (never combined with sys/hip/hsa tracing)
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  NAME=$(echo "$SET" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc_$NAME" -o pmc -- $BENCH > "$OUT/pmc_$NAME.log" 2>&1 || { echo "pmc pass $SET failed"; tail -5 "$OUT/pmc_$NAME.log"; }
done
STATS=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp "$STATS" "$OUT/${TAG}_kernel_stats.csv"
python3 $ROOT/tools/pmc_summary.py "$OUT" "$OUT/${TAG}_pmc_summary.json" "$BENCH"
