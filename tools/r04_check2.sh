cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 1100 python tools/r04_rcp_ab.py > gpurun_out/r04/rcp_ab.txt 2>&1; rc=$?
tail -40 gpurun_out/r04/rcp_ab.txt
exit $rc
