#!/bin/bash
cd "$(dirname "$0")/.."
for k in 20 50; do
RPT_BENCH_VERBOSE=1 python bench.py --steps $k --warmup 5 --no-cpu-baseline 2>gpurun_out/err.txt | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print($k, d['value'], d['ms_per_step'], d['roofline']['frac'])"
grep "blocking launch" gpurun_out/err.txt
done
