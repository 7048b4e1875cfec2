#!/bin/bash
# does TD_TD_BUSY follow the dwords returned?  kernel 41 (packed count + late id) against arm 625 (packed count only) and arm 273 (neither)
cd "$(dirname "$0")/.."
for v in 41 625 273; do
  echo "== bunny 4K, variant $v"
  RPT_ABLATE_ARGS="--diag" bash tools/pmc_scene.sh r03_tdcheck_$v bunny $v "TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" 2>&1 | grep -v -E "amdgpu.ids|^pass"
done | tee gpurun_out/r03_td_check.txt
