#!/bin/bash
# first GPU pass of round 3: parity of the persistent kernels (60 staged, 62 direct), then A/B against 41 / 43
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_frame_matches_oracle and (60 or 62)" > gpurun_out/r03_first_parity.log 2>&1 || { tail -30 gpurun_out/r03_first_parity.log; exit 1; }
tail -3 gpurun_out/r03_first_parity.log
timeout -k 10 400 python tools/configs.py --variants 41,43,60,62,41,60 --only bunny,shadows --frames 30 > gpurun_out/r03_first_ab.log 2>&1 || { tail -30 gpurun_out/r03_first_ab.log; exit 1; }
cat gpurun_out/r03_first_ab.log | grep -v '^\['
