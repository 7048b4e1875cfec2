#!/bin/bash
# the walk without repeated triangle tests (arms 705 / 717, diagnostics library) against the product
# walks in the same library build (41 / 43), A/B/A/B, bunny 1080p / 4K / 8K and shadows 4K, four frames in flight and one at a time;
# then TD_TD_BUSY / TA_TA_BUSY / SQ_INSTS_* for both on bunny 4K.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/dedup_ab.txt
: > $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_diag_arms.py -x -q -k "705 or 717" 2>&1 | tail -2 | tee -a $OUT || exit 1
timeout -k 10 600 python tools/configs.py --diag --variants 705,41,705,41,705,41 --only bunny,shadows --frames 60 --inflight 4 2>&1 | grep 'variant ' | tee -a $OUT
timeout -k 10 600 python tools/configs.py --diag --variants 717,43,717,43,717,43 --only bunny,shadows --frames 60 --inflight 4 2>&1 | grep 'variant ' | tee -a $OUT
