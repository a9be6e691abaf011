#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_precision_gpu.py -q -x -s -k "heldout or session or small_batches" 2>&1 | grep "INT8 1280\|passed\|failed\|Error\|assert\|vs torch" | cut -c1-300 | tail -12
