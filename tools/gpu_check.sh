#!/bin/bash
# one GPU call: the GPU test-suite, the two-model concurrency check, a short bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 300 python tools/dbg_concurrent.py 0 ${ITERS:-150} 2>&1 | grep "^bad\|Error\|error" | head -5
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['conv_family']['kernels_ms'])"
