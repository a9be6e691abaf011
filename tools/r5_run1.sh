#!/bin/bash
# round-5 GPU call 1: the new native-sharded tests, the multi-frame calibration experiment, the default bench, then the whole suite
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_native_gpu.py -x -q -s 2>&1 | tail -25 > gpurun_out/r05a_sharded.txt; tail -5 gpurun_out/r05a_sharded.txt
timeout 600 python tools/q8_multi.py 640 480 --ks 1,4,8,16 > gpurun_out/r05a_q8_multi_480.txt 2>&1; cat gpurun_out/r05a_q8_multi_480.txt | tail -8
timeout 600 python tools/q8_multi.py 1280 720 --ks 1,8 > gpurun_out/r05a_q8_multi_720.txt 2>&1; cat gpurun_out/r05a_q8_multi_720.txt | tail -4
timeout 600 python bench.py 2>gpurun_out/r05a_bench_err.txt | tail -1 > gpurun_out/r05a_bench_default.json
python -c "import json; d=json.load(open('gpurun_out/r05a_bench_default.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05a_gputests.txt; tail -5 gpurun_out/r05a_gputests.txt
