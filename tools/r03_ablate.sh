#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  echo "== variant $v"
  timeout -k 10 120 python tools/ablate.py --variant $v --only empty,sphere_light,bunny_mesh_only,bunny,shadows 2>&1 | grep -v amdgpu.ids | grep -v '^{'
done
