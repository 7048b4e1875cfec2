#!/bin/bash
# After a change of librpt_hip.so: the evidence of record again, on ONE build, in one call on the GPU box:
#   gpurun --timeout 1200 -- 'bash tools/r04_refresh_record.sh'
# then HERE:  python tools/collect_profiles.py r04_... ; cp gpurun_out/r04_bench_*.json gpurun_out/r04_configs_default.txt profiles/ ;
#             python tools/table_of_record.py r04 --write
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r04
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_suite.log 2>&1 || { tail -20 gpurun_out/r04/gpu_suite.log; exit 1; }
tail -2 gpurun_out/r04/gpu_suite.log
bash tools/r04_final_profiles.sh || exit 1
for t in r04_bunny_3840x2160 r04_bunny_1920x1080 r04_shadows_3840x2160 r04_arch_1920x1080 r04_cube_640x480 r04_bunny_7680x4320 r04_cubes_3840x2160; do
  cp gpurun_out/prof_$t/${t}_pmc_summary.json profiles/ || exit 1      # (on the box: so that the bench lines below find this build's traffic)
done
bash tools/r04_bench_lines.sh
