#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( HELD=8 timeout 600 python tools/q8_multi.py 640 480 --ks 16,32,48
  HELD=8 timeout 900 python tools/q8_multi.py 1280 720 --ks 16,32 ) > gpurun_out/r05g_q8_k32.txt 2>&1
cat gpurun_out/r05g_q8_k32.txt
