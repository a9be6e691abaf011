#!/bin/bash
# stability: the GPU suite twice in a row on one box (no -x: every failure is listed)
mkdir -p gpurun_out
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/suite_run$i.txt
  tail -3 gpurun_out/suite_run$i.txt
done
