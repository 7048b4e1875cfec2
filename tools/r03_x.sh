#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_gpu_diag_arms.py -x -q -k "529 or 541" 2>&1 | tail -2
timeout -k 10 400 python tools/configs.py --diag --variants 273,529,285,541,273,529,285,541 --only bunny,shadows --frames 40 2>&1 | grep 'variant '
