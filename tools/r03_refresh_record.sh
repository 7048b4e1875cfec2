#!/bin/bash
# After a change of librpt_hip.so: the evidence of record again, on one build.  Run HERE (it calls gpurun twice):
#   1. traces + PMC passes + bench lines + default-path table + walk A/B      (tools/r03_final_profiles.sh on the box)
#   2. condensed summaries into profiles/, committed so that bench.py finds this build's HBM traffic
#   3. bench lines once more (now with roofline.traffic)                        (tools/r03_bench_lines.sh on the box)
#   4. DESIGN.md 6.1 / 6.2 and profiles/README.md regenerated                   (tools/table_of_record.py --write)
set -e
cd "$(dirname "$0")/.."
/usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r03_final_profiles.sh'
python tools/collect_profiles.py r03_bunny_3840x2160 r03_bunny_1920x1080 r03_shadows_3840x2160 r03_arch_1920x1080 r03_cube_640x480 r03_bunny_7680x4320 r03_cubes_3840x2160
/usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/r03_bench_lines.sh'
cp gpurun_out/r03_bench_*x*.json profiles/
cp gpurun_out/r03_configs_default.txt profiles/
cp gpurun_out/r03_walk_async_ab.txt profiles/r03_walk_async_ab_final.txt
cp gpurun_out/r03_walk_blocking_ab.txt profiles/r03_walk_blocking_ab_final.txt
python tools/table_of_record.py r03 --write
