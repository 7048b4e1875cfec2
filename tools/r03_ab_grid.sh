#!/bin/bash
# the root table (descend_from_root): 41 against 273 (serial descent, count from its own field), the latency walk (573) against 605
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
timeout -k 10 300 python -m pytest tests/test_gpu_diag_arms.py -x -q -k "605 or 573 or 273" 2>&1 | tail -2
timeout -k 10 500 python tools/configs.py --diag --variants 273,41,273,41,273,41 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
timeout -k 10 500 python tools/configs.py --diag --variants 605,573,605,573,605,573 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
