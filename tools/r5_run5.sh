#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 400 python tools/q8_multi.py 640 480 --ks 1,8,16
  for o in 2,0,0 1,1,1 0,0,0; do echo "== opts $o"; timeout 300 python tools/q8_multi.py 640 480 --ks 8,16 --opts $o; done
  timeout 400 python tools/q8_multi.py 1280 720 --ks 8,16
  timeout 400 python tools/q8_multi.py 1280 720 --ks 8 --untextured ) > gpurun_out/r05e_q8_efr.txt 2>&1
cat gpurun_out/r05e_q8_efr.txt
timeout 300 python -m pytest tests/test_properties_gpu.py -q -s -k "keep_serving" 2>&1 | grep "Track latency\|passed\|failed"
