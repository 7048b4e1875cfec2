# round 4, first GPU call: the suite on the certified bounds + derived shadow culls, smoke, the bench lines that matter
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_suite.log 2>&1; rc=$?
tail -15 gpurun_out/r04/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 &&
for wl in bunny shadows; do
  python bench.py --steps 50 --warmup 10 --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04/bench_$wl.json &&
  python -c "import json;d=json.load(open('gpurun_out/r04/bench_$wl.json'));r=d['roofline'];print('$wl',d['value'],d['ms_per_step'],r['frac'],r.get('launch_ms'),d.get('value_blocking'),d.get('ms_per_frame_blocking'))"
done
