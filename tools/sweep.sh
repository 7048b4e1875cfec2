#!/bin/bash
# resolution sweep + ablation on the current library
OUT=gpurun_out/sweep_final.txt
echo "# tools/inflight.py --world 1 --inflight 1,3 --frames 40 per scene and size (round's last library: non-temporal stores)" > $OUT
for S in bunny shadows; do
  for WH in "640 360" "1280 720" "1920 1080" "2560 1440" "3840 2160" "7680 4320" "15360 8640"; do
    set -- $WH
    python tools/inflight.py --scene $S --width $1 --height $2 --world 1 --inflight 1,3 --frames 40 2>&1 | grep -v amdgpu.ids | sed "s/^/$S ${1}x${2}: /" >> $OUT
  done
done
python tools/ablate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ablation_final.txt
cat $OUT gpurun_out/ablation_final.txt
