#!/bin/bash
# end-of-round evidence on the final library: rocprofv3 trace + PMC passes per BASELINE configuration (one block per kernel in the
# summaries: tools/pmc_summary.py), the default-path table over all eleven scene/size pairs
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
prof() { tag=$1; shift; bash tools/profile.sh $tag "$@" > gpurun_out/prof_$tag.log 2>&1; echo "profiled $tag"; }
prof r04_bunny_3840x2160
prof r04_bunny_1920x1080 --width 1920 --height 1080
prof r04_shadows_3840x2160 --workload shadows
prof r04_arch_1920x1080 --workload arch --width 1920 --height 1080
prof r04_cube_640x480 --workload cube --width 640 --height 480
prof r04_bunny_7680x4320 --width 7680 --height 4320
prof r04_cubes_3840x2160 --workload cubes
python tools/configs.py --variants 0 --frames 60 --inflight 4 2>&1 | grep 'variant ' > gpurun_out/r04_configs_default.txt; cat gpurun_out/r04_configs_default.txt
