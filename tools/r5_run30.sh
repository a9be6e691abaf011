#!/bin/bash
# round-5 evidence of the final code state: GPU suite, smoke, default bench line, Track timeline, shard timing, rocprofv3 stats + PMC
# passes of the headline command and of the INT8 720p command
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -12 > gpurun_out/r05_gputests.txt; tail -4 gpurun_out/r05_gputests.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>gpurun_out/r05_bench_err.txt | tail -1 > gpurun_out/r05_bench_default.json
python -c "import json; d=json.load(open('gpurun_out/r05_bench_default.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['int8_720p']['value'], d['track']['value'], d['track_int8']['value'])"
python tools/track_timeline.py > gpurun_out/r05_track_timeline.txt 2>&1; tail -2 gpurun_out/r05_track_timeline.txt
python tools/time_shard.py > gpurun_out/r05_time_shard.txt 2>&1; head -3 gpurun_out/r05_time_shard.txt
timeout 900 bash tools/profile_round.sh r05_register_n252 2>&1 | tail -6 | cut -c1-160
timeout 1200 bash tools/profile_round.sh r05_register_int8_720p --dtype int8 --width 1280 --height 720 2>&1 | tail -8 | cut -c1-160
ls -la gpurun_out | head -40
