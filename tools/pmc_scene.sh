#!/bin/bash
# PMC passes over tools/ablate.py for one scene/variant. usage: bash tools/pmc_scene.sh <tag> <scene> <variant> "<counters pass 1>" "<pass 2>" ...
set -o pipefail
TAG=$1; SCENE=$2; VAR=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  # (a counter set the hardware cannot collect in one pass makes rocprofv3 abort and then wait for ever: bounded, and reported)
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o pmc -- python3 $ROOT/tools/ablate.py --only $SCENE --variant $VAR --frames 5 $RPT_ABLATE_ARGS > "$OUT/p$i.log" 2>&1 || { echo "pass $i ($SET) failed"; grep -m1 -E "exceeds|error code" "$OUT/p$i.log"; }
  echo "pass $i done"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        if 'rpt_render' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in sorted(agg.items()): print(f"{k:36s} mean={sum(v)/len(v):.6g} n={len(v)}")
PY
