#!/bin/bash
# A/B: left-over rows on conv_smallx_kernel
mkdir -p gpurun_out
python tools/ab_wall.py fpt_set_rem_smallx 0 1 0 1 > gpurun_out/ab_rem_smallx.txt 2>&1
python tools/profile_shard.py 252 fpt_set_rem_smallx 1 > gpurun_out/shard_252_x.txt 2>&1
python -m pytest tests/test_nn_gpu.py -x -q -m gpu > gpurun_out/t_conv.txt 2>&1
cat gpurun_out/ab_rem_smallx.txt; grep "conv_512\|conv_b2\|slice" gpurun_out/shard_252_x.txt; tail -3 gpurun_out/t_conv.txt
