#!/bin/bash
# fifth soak, shipped library: animated sweeps at 1920x1080 along four more paths (every state: kernels 41 and 43 against the un-culled kernel)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r03_soak5_sweeps.txt
for p in 14 15 16 17; do
  echo "== path $p, 12000 states, 1920x1080" >> gpurun_out/r03_soak5_sweeps.txt
  timeout -k 10 280 python tools/verify_sweep.py --states 12000 --width 1920 --height 1080 --path $p 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_soak5_sweeps.txt || exit 1
  tail -1 gpurun_out/r03_soak5_sweeps.txt
done
