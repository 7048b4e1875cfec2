#!/bin/bash
# the helper threads of rpt_set_objects (csrc/rpt_workers.hpp): host time per animated frame with and without them.
# At 4K the animated cubes sequence is device-bound (its frames cost more than the still one at t = 3); at 1080p the host is what limits.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for size in "1920 1080" "3840 2160"; do for n in 0 4 0 4; do
  echo "== RPT_HOST_THREADS=$n  $size"
  RPT_HOST_PROFILE=1 RPT_HOST_THREADS=$n python tools/host_cost.py cubes $size 2>&1 | grep -v amdgpu.ids
done; done 2>&1 | tee gpurun_out/r03_host_threads.txt
