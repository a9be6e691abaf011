#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_precision_gpu.py -q -s -k "clear_one" 2>&1 | grep "held-out scene\|passed\|failed\|Error" | cut -c1-250
