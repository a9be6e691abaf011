#!/bin/bash
# round-6 session script (scratch): the copy-command-free Register serving path (host frames / masks, frame record, read-back)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r6c_tests.txt
timeout 200 python tools/time_register_host.py > gpurun_out/r6c_host.txt 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6c_bench.json
timeout 600 python tools/q8_blocks.py > gpurun_out/r06_q8_blocks_int8.log 2>&1
tail -6 gpurun_out/r6c_tests.txt; cat gpurun_out/r6c_host.txt | tail -3
python -c "
import json; d=json.load(open('gpurun_out/r6c_bench.json')); print('bench', d['value'], d['ms_per_step'], 'host', d['host_frame']['ms_per_step'], 'track', d['track']['value'], d['track']['host_frame_value'])"
