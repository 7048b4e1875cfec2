#!/bin/bash
cd "$(dirname "$0")/.."
echo "== default"; timeout -k 10 200 python tools/configs.py --variants 41,63 --only bunny --frames 30 2>&1 | grep 'variant '
echo "== RPT_NO_BAND=1"; RPT_NO_BAND=1 timeout -k 10 200 python tools/configs.py --variants 41,63 --only bunny --frames 30 2>&1 | grep 'variant '
