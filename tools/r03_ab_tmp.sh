#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_gpu_diag_arms.py -x -q -k "689 or 701" 2>&1 | tail -2
timeout -k 10 500 python tools/configs.py --diag --variants 41,689,41,689,41,689 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
timeout -k 10 500 python tools/configs.py --diag --variants 43,701,43,701,43,701 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
