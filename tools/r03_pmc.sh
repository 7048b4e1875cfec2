#!/bin/bash
# usage: r03_pmc.sh <scene> <variant> ...   PMC counters of tools/ablate.py per variant
cd "$(dirname "$0")/.."
SCENE=$1; shift
for v in "$@"; do
  echo "== $SCENE variant $v"
  bash tools/pmc_scene.sh r03_${SCENE}_v$v $SCENE $v "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" 2>&1 | grep -v amdgpu.ids
done
