#!/bin/bash
# one wave per workgroup (the product kernels) against four (arms 689 / 701: the same walks, launched as 32 x 8 strips)
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 500 python tools/configs.py --diag --variants 689,41,689,41,689,41 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
timeout -k 10 500 python tools/configs.py --diag --variants 701,43,701,43,701,43 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
timeout -k 10 300 python tools/configs.py --variants 0 --frames 60 --inflight 4 2>&1 | grep 'variant '
