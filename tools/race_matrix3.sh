#!/bin/bash
mkdir -p gpurun_out/race
for v in ${VARS:-0}; do
  FP_VERTEX_VAR=$v FP_DISABLE_GPU_LOCK=1 timeout 120 python tools/dbg_concurrent3.py register ${ITERS:-150} > gpurun_out/race/var_$v.log 2>&1
  echo "VAR $v: $(tail -1 gpurun_out/race/var_$v.log)"
done
