#!/bin/bash
# usage: r03_ab.sh "<variants>" "<scenes>" [frames] [label]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "test_frame_matches_oracle and (305 or 337 or 401 or 285 or 317 or 349)" 2>&1 | tail -2
timeout -k 10 500 python tools/configs.py --variants "$1" --only "$2" --frames ${3:-30} 2>&1 | grep -v amdgpu.ids | grep -v '^\[' | tee gpurun_out/r03_ab_${4:-last}.log
