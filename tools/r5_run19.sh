#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ab_wall.py fpt_set_att_tail 0 1 0 1 0 1 2>&1 | tail -6
timeout 600 python tools/ab_pipeline.py fpt_set_att_tail 0 1 0 1 2>&1 | tail -4 | cut -c1-300
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_discriminative_gpu.py tests/test_golden_gpu.py tests/test_precision_gpu.py -q -x -k "attention or refiner or scorer or register or golden or Register" 2>&1 | grep "passed\|failed" | tail -3
