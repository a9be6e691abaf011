#!/bin/bash
# round-6 session script (scratch): validation of the stage-mask trunk, the restructured configs[4] tests, the self-launching bench
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_precision_gpu.py -m gpu -q -x -k "configs4 or experimental_level or holds_95 or fp8_meets or small_batches or needs_calibration or calibration_session" 2>&1 | tail -15 > gpurun_out/r6a_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6a_smoke.txt 2>&1
FP_BENCH_FORCE_SHARD=1 timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-extras > gpurun_out/r6a_bench_shard1.json 2> gpurun_out/r6a_bench_shard1.err
timeout 120 python bench.py --gpus 2 --steps 2 > gpurun_out/r6a_bench_gpus2.out 2> gpurun_out/r6a_bench_gpus2.err; echo "rc=$?" >> gpurun_out/r6a_bench_gpus2.err
timeout 600 python bench.py --steps 10 --warmup 3 2> gpurun_out/r6a_bench_default.err | tail -1 > gpurun_out/r6a_bench_default.json
timeout 500 python tools/q8_blocks.py --prec fp8 --masks 4,2,8,15 --held 4 > gpurun_out/q8_blocks_fp8.log 2>&1
tail -5 gpurun_out/r6a_tests.txt; cat gpurun_out/r6a_smoke.txt | tail -2; tail -c 600 gpurun_out/r6a_bench_shard1.json; tail -3 gpurun_out/r6a_bench_shard1.err; cat gpurun_out/r6a_bench_gpus2.err | tail -3
