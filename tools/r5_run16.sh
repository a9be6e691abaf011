#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_precision_gpu.py -q -s -k "heldout" 2>&1 | grep "held-out scene\|INT8 1280\|passed\|failed\|Error\|assert" | cut -c1-330
