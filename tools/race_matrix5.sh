#!/bin/bash
mkdir -p gpurun_out/race
for l in ${LIBS:-noslp_noprio}; do
  FP_LIB_PATH=$PWD/tools/_bin/lib_$l.so FP_DISABLE_GPU_LOCK=1 timeout 400 python tools/dbg_concurrent3.py register ${ITERS:-1000} > gpurun_out/race/lib_$l.log 2>&1
  echo "$l: $(tail -1 gpurun_out/race/lib_$l.log)"
  FP_LIB_PATH=$PWD/tools/_bin/lib_$l.so FP_DISABLE_GPU_LOCK=1 timeout 600 python tools/dbg_concurrent.py 0 ${ITERS2:-500} > gpurun_out/race/full_$l.log 2>&1
  echo "$l full: $(grep '^bad' gpurun_out/race/full_$l.log)"
done
