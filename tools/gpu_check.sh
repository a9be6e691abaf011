#!/bin/bash
# one GPU call for a round's evidence: the GPU test-suite, smoke(), the default bench line, the Track graph timeline, and
# (tools/profile_round.sh) kernel-trace stats + FETCH / WRITE / MFMA counter passes of the headline command
# usage: tools/gpu_check.sh <tag>   -> gpurun_out/<tag>_*   (copy what you keep into profiles/)
TAG=${1:-check}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/${TAG}_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_default.json
python -c "import sys,json; d=json.load(open('gpurun_out/${TAG}_bench_default.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv_family']['kernels_ms']); print('track', d['track']['value'], d['track']['host_frame_value'], d['track_int8']['value'])"
bash tools/profile_track.sh 2>&1 | tail -3
python tools/track_timeline.py > gpurun_out/${TAG}_track_timeline.txt 2>&1; tail -3 gpurun_out/${TAG}_track_timeline.txt
timeout 900 bash tools/profile_round.sh ${TAG}_register_n252 2>&1 | tail -8
