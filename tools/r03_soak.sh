#!/bin/bash
# end-of-round soak of rpt_verify_frame on the final library: fresh seed ranges of every generator
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 330 python tools/verify_fuzz.py --first 300000 --last 335000 --kinds walls 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak_walls.txt | tail -3 &&
timeout -k 10 330 python tools/verify_fuzz.py --first 300000 --last 335000 --kinds ellipsoids 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak_ellipsoids.txt | tail -3 &&
timeout -k 10 330 python tools/verify_fuzz.py --first 108000 --last 136000 --kinds meshwalls 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak_meshwalls.txt | tail -3 &&
timeout -k 10 200 python tools/verify_fuzz.py --first 42000 --last 48000 --kinds random,extreme,close 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_soak_fuzz.txt | tail -3
