# the default bench command with roofline.valu in the line (library of record: the PMC summaries' hash must match)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 600 python bench.py > gpurun_out/r04/bench_valu_default.json 2> gpurun_out/r04/bench_valu_default.err; rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/bench_valu_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["traffic"])
print("alone    ", r.get("valu"))
print("in flight", r["device_in_flight"].get("valu"))
PY
exit $rc
