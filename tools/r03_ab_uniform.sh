#!/bin/bash
# UNIFORM: node and triangle records through the scalar cache where the whole wave stands in one node: 657 against 41, 669 against 43
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_gpu_diag_arms.py -x -q -k "657 or 669" 2>&1 | tail -2
timeout -k 10 500 python tools/configs.py --diag --variants 41,657,41,657,41,657 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
timeout -k 10 500 python tools/configs.py --diag --variants 43,669,43,669,43,669 --only bunny,shadows --frames 60 2>&1 | grep 'variant '
