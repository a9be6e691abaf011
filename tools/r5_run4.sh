#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -60 > gpurun_out/r05d_gputests.txt; tail -40 gpurun_out/r05d_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
