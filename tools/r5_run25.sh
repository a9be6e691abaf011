#!/bin/bash
# INT8 held-out accuracy against the activation headroom (scale = |max| * headroom / 255)
mkdir -p gpurun_out
for h in 1.0 1.25 1.6 2.0; do
  HELD=8 python tools/q8_multi.py --ks 16 --headroom $h 2>&1 | grep "K=16" | sed "s/^/headroom $h: /" >> gpurun_out/q8_headroom.txt
done
cat gpurun_out/q8_headroom.txt
