#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
