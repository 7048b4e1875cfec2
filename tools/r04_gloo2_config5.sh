#!/bin/bash
# rehearsal: `python bench.py --gpus 2` BARE over gloo (both ranks on this box's one GPU) with the full-size config 5 block (7680x4320)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r04
RPT_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_gloo2_config5.json 2> gpurun_out/r04/bench_gloo2_config5.err
echo rc=$?
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_gloo2_config5.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["config"]["sharding"])
print(json.dumps(d["comm"]["config5"], indent=1))
print(d["comm"]["split_timings_ms_per_frame"])
PY
tail -3 gpurun_out/r04/bench_gloo2_config5.err
