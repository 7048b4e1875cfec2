#!/bin/bash
cd "$(dirname "$0")/.."
for st in 0 1 0 1; do for k in 20 200; do
RPT_STAGGER=$st python bench.py --steps $k --warmup 5 --no-cpu-baseline --check 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('stagger',$st, $k, d['value'], d['ms_per_step'], d['roofline']['frac'], d['check'])"
done; done
for st in 0 1 0 1; do
RPT_STAGGER=$st python tools/configs.py --inflight 4 --variants 0 --only bunny,shadows --frames 60 2>&1 | grep 'variant ' | sed "s/^/stagger $st /"
done
