#!/bin/bash
# second end-of-round soak on the final library: animated sweeps along six new paths, more meshwalls seeds
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r03_soak2_sweeps.txt
for p in 6 7 8 9 10 11; do
  echo "== path $p, 20000 states, 1280x544" >> gpurun_out/r03_soak2_sweeps.txt
  timeout -k 10 280 python tools/verify_sweep.py --states 20000 --width 1280 --height 544 --path $p 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03_soak2_sweeps.txt || exit 1
  tail -1 gpurun_out/r03_soak2_sweeps.txt
done
timeout -k 10 500 python tools/verify_fuzz.py --first 40000 --last 80000 --kinds meshwalls 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak2_meshwalls.txt | tail -2
