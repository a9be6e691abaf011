#!/bin/bash
# round-6 session script (scratch): the round's evidence in one call (gpu_check.sh) + per-rank shard times
bash tools/gpu_check.sh r06 > gpurun_out/r06_gpu_check.log 2>&1
timeout 300 python tools/time_shard.py > gpurun_out/r06_time_shard.txt 2>&1
tail -30 gpurun_out/r06_gpu_check.log; tail -8 gpurun_out/r06_time_shard.txt
