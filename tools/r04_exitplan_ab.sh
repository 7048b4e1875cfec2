# the packed exit faces (make exitfaces): the GPU suite on the experiment library first, then the A/B against the product library
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
RPT_HIP_LIB=$PWD/relativitypathtracer_amd/librpt_hip_exitfaces.so timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/exitplan_suite.log 2>&1; rc=$?
tail -4 gpurun_out/r04/exitplan_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tools/r04_exitplan_ab.py > gpurun_out/r04/exitplan_ab.txt 2>&1; rc=$?
grep "=>" -B0 gpurun_out/r04/exitplan_ab.txt | tail -20
exit $rc
