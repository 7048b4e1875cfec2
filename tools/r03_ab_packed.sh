#!/bin/bash
# the leaf count packed into the node record's begin word (two 16-B loads per node visit instead of three instructions):
# 41 (packed) against 273 (the same walk, count from its own field), 573 (packed latency walk) against 575
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_gpu_diag_arms.py -x -q -k "575 or 573" 2>&1 | tail -2
timeout -k 10 500 python tools/configs.py --diag --variants 575,573,575,573,575,573 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
