#!/bin/bash
# shard profile: per-kernel tables at 32 / 63 / 252
mkdir -p gpurun_out
for n in 32 63 252; do python tools/profile_shard.py $n > gpurun_out/shard_$n.txt 2>&1; done
tail -30 gpurun_out/shard_32.txt
