#!/bin/bash
# one GPU call: the GPU test-suite, smoke(), the default bench line, and a kernel-trace stats pass of the bench command
# usage: tools/gpu_check.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-check}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/${TAG}_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_default.json
python -c "import sys,json; d=json.load(open('gpurun_out/${TAG}_bench_default.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv_family']['kernels_ms'])"
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-mfma-peak --no-extras > $OUT/stats.log 2>&1)
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cp "$S" gpurun_out/${TAG}_register_n252_kernel_stats.csv && head -8 gpurun_out/${TAG}_register_n252_kernel_stats.csv
