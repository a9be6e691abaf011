#!/bin/bash
mkdir -p gpurun_out/race
FP_DISABLE_GPU_LOCK=1 timeout 600 python tools/dbg_concurrent.py 0 ${ITERS:-300} > gpurun_out/race/full_nolock.log 2>&1
tail -4 gpurun_out/race/full_nolock.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline'])"
