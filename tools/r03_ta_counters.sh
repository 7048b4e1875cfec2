#!/bin/bash
# how busy are the texture-address / data units during a frame?  (one launch at a time under the profiler; TA_TA_BUSY_sum and
# TD_TD_BUSY_sum are summed over the chip's 256 CUs, GRBM_GUI_ACTIVE over its 8 XCDs)
cd "$(dirname "$0")/.."
SETS=("TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum")
run() { tag=$1; scene=$2; v=$3; shift 3; echo "== $tag"; RPT_ABLATE_ARGS="$*" bash tools/pmc_scene.sh r03_ta_$tag $scene $v "${SETS[@]}" 2>&1 | grep -v -E "amdgpu.ids|^pass"; }
{
run bunny4k_41 bunny 41
run bunny4k_43 bunny 43
run bunny8k_41 bunny 41 --width 7680 --height 4320
run bunny1080_43 bunny 43 --width 1920 --height 1080
run shadows4k_41 shadows 41
run arch1080_44 arch 44 --width 1920 --height 1080
run cubes4k_44 cubes 44
} | tee gpurun_out/r03_ta_counters.txt
