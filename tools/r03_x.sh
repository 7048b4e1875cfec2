#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_gpu_diag_arms.py -x -q -k "2573" 2>&1 | tail -2
timeout -k 10 400 python tools/configs.py --diag --variants 573,2573,573,2573,573,2573 --only bunny,shadows --frames 40 2>&1 | grep 'variant '
