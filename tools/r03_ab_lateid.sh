#!/bin/bash
# the triangle id read once at the end of a walk instead of with every record tested (LATE_ID): 625 against 41, 637 against 43
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_gpu_diag_arms.py -x -q -k "625 or 637" 2>&1 | tail -2
timeout -k 10 500 python tools/configs.py --diag --variants 41,625,41,625,41,625 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
timeout -k 10 500 python tools/configs.py --diag --variants 43,637,43,637,43,637 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
