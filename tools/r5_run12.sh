#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_precision_gpu.py -q -s -k "heldout or fp8_meets or clear_one or session_api or reproducible or small_batches or needs_calibration or per_layer" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > gpurun_out/r05l_q8_tests.txt; tail -70 gpurun_out/r05l_q8_tests.txt | cut -c1-420
