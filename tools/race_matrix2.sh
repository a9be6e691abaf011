#!/bin/bash
mkdir -p gpurun_out/race
for m in ${MODES:-register track}; do
  FP_DISABLE_GPU_LOCK=1 timeout 300 python tools/dbg_concurrent3.py $m ${ITERS:-100} > gpurun_out/race/agg_$m.log 2>&1
  tail -${TAIL:-30} gpurun_out/race/agg_$m.log
done
