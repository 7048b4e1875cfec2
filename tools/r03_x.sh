#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 500 python tools/configs.py --inflight 4 --variants 41,43,41,43 --only bunny,shadows --sizes 1280x720,1920x1080,2560x1440,3200x1800,3840x2160 --frames 40 2>&1 | grep 'variant '
