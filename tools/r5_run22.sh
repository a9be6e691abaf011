#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -12 > gpurun_out/r05_gputests.txt; tail -6 gpurun_out/r05_gputests.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>gpurun_out/r05_bench_err.txt | tail -1 > gpurun_out/r05_bench_default.json
python -c "import json; d=json.load(open('gpurun_out/r05_bench_default.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['int8_720p']['value'], d['track']['value'], d['track_int8']['value'])"
