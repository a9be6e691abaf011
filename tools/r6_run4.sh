#!/bin/bash
# round-6 session script (scratch): final evidence of the round
bash tools/gpu_check.sh r06 > gpurun_out/r06_gpu_check.log 2>&1
timeout 300 python tools/time_shard.py > gpurun_out/r06_time_shard.txt 2>&1
FP_BENCH_FORCE_SHARD=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras 2>gpurun_out/r06_bench_shard_world1.err | tail -1 > gpurun_out/r06_bench_shard_world1.json
timeout 600 bash tools/profile_round.sh r06_register_f16_720p --width 1280 --height 720 > gpurun_out/r06_profile_720p.log 2>&1
tail -25 gpurun_out/r06_gpu_check.log | cut -c1-220; tail -4 gpurun_out/r06_time_shard.txt; tail -c 400 gpurun_out/r06_bench_shard_world1.json; tail -3 gpurun_out/r06_bench_shard_world1.err; tail -5 gpurun_out/r06_profile_720p.log | cut -c1-200
