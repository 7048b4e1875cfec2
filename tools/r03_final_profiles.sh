#!/bin/bash
# end-of-round evidence on the final library: rocprofv3 trace + PMC passes per BASELINE configuration, bench lines with --check, the
# default-path table over all eleven scene/size pairs, the walk A/B against round 2's walk (diagnostics library)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
prof() { tag=$1; shift; bash tools/profile.sh $tag "$@" > gpurun_out/prof_$tag.log 2>&1; echo "profiled $tag"; }
prof r03_bunny_3840x2160
prof r03_bunny_1920x1080 --width 1920 --height 1080
prof r03_shadows_3840x2160 --workload shadows
prof r03_arch_1920x1080 --workload arch --width 1920 --height 1080
prof r03_cube_640x480 --workload cube --width 640 --height 480
prof r03_bunny_7680x4320 --width 7680 --height 4320
prof r03_cubes_3840x2160 --workload cubes
bench() { name=$1; shift; python bench.py --steps 50 --warmup 5 --check "$@" 2>/dev/null | tail -1 > gpurun_out/r03_bench_$name.json; echo "bench $name: $(python -c "import json;d=json.load(open('gpurun_out/r03_bench_$name.json'));print(d['value'],d['ms_per_step'],d['ms_per_frame_blocking'],d['roofline']['frac'],d['roofline']['device_in_flight']['frac'],d['check'])")"; }
bench bunny_3840x2160
bench bunny_1920x1080 --width 1920 --height 1080
bench shadows_3840x2160 --workload shadows
bench arch_1920x1080 --workload arch --width 1920 --height 1080
bench cube_640x480 --workload cube --width 640 --height 480
bench bunny_7680x4320 --width 7680 --height 4320
bench cubes_3840x2160 --workload cubes
python tools/configs.py --variants 0 --frames 60 2>&1 | grep 'variant ' > gpurun_out/r03_configs_default.txt; cat gpurun_out/r03_configs_default.txt
python tools/configs.py --diag --variants 141,41,141,41 --only bunny,shadows --frames 60 2>&1 | grep 'variant ' > gpurun_out/r03_walk_async_ab.txt
python tools/configs.py --diag --variants 143,43,143,43 --only bunny,shadows --frames 60 2>&1 | grep 'variant ' > gpurun_out/r03_walk_blocking_ab.txt
cat gpurun_out/r03_walk_async_ab.txt gpurun_out/r03_walk_blocking_ab.txt
