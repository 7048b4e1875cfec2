#!/bin/bash
# host time per exchanged frame of bench.py's N > 1 path, one rank over RCCL (RPT_FORCE_DIST): native ncclGather (ctypes) vs torch.distributed.gather
cd "$(dirname "$0")/.."
for ex in native torch native torch; do for fpe in 1 4; do
  echo "== RPT_EXCHANGE=$ex frames-per-exchange=$fpe"
  RPT_EXCHANGE=$ex RPT_FORCE_DIST=1 RPT_SPLIT=equal RPT_BENCH_VERBOSE=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --frames-per-exchange $fpe 2>&1 | grep -E '^\[bench\]|"ms_per_step"' | sed -E 's/.*("ms_per_step": [0-9.]+).*/\1/'
done; done
