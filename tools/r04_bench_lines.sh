#!/bin/bash
# the bench lines of record (after the PMC summaries of THIS build have been committed to profiles/: roofline.traffic is then filled in)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
bench() { name=$1; shift; python bench.py --steps 50 --warmup 5 --check "$@" 2>/dev/null | tail -1 > gpurun_out/r04_bench_$name.json; echo "bench $name: $(python -c "import json;d=json.load(open('gpurun_out/r04_bench_$name.json'));r=d['roofline'];print(d['value'],d['ms_per_step'],d['ms_per_step_median_of_batches'],d['ms_per_frame_blocking'],r['launch_ms'],r['frac'],r['device_in_flight']['frac'],r['traffic'],r['device_in_flight']['traffic'],d['animated']['ms_per_step'],d['cpu_baseline']['value'] if 'cpu_baseline' in d else None,d['cpu_baseline'].get('threads_used') if 'cpu_baseline' in d else None,d['check'])")"; }
bench bunny_3840x2160
bench bunny_1920x1080 --width 1920 --height 1080
bench shadows_3840x2160 --workload shadows
bench arch_1920x1080 --workload arch --width 1920 --height 1080
bench cube_640x480 --workload cube --width 640 --height 480
bench bunny_7680x4320 --width 7680 --height 4320
bench cubes_3840x2160 --workload cubes
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_bench_driver_form.json; python -c "import json;d=json.load(open('gpurun_out/r04_bench_driver_form.json'));print('driver form (20 steps):',d['value'],d['ms_per_step'],d['ms_per_step_median_of_batches'],d['ms_per_step_batches'])"
