#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ab_wall.py fpt_set_rem_fork 0 1 0 1 2>&1 | tail -6
timeout 600 python tools/ab_wall.py fpt_set_halo_wreg 0 1 2 0 1 2 2>&1 | tail -8
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_discriminative_gpu.py tests/test_golden_gpu.py -q -x 2>&1 | tail -3
