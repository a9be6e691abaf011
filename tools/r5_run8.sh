#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( for wq in 1,2 0,2 1,1; do echo "== wq $wq"; HELD=8 timeout 600 python tools/q8_multi.py 640 480 --ks 8,16,32 --wq $wq; done
  echo "== 720p wq 1,2"; HELD=8 timeout 900 python tools/q8_multi.py 1280 720 --ks 16,32 --wq 1,2 ) > gpurun_out/r05h_q8_tapsum.txt 2>&1
cat gpurun_out/r05h_q8_tapsum.txt
