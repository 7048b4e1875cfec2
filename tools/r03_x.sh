#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 300 python tools/configs.py --variants 273,785,273,785 --only bunny,shadows --frames 30 2>&1 | grep 'variant '
