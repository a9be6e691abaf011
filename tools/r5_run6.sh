#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( for wq in 1,1 0,1; do echo "== wq $wq"; timeout 400 python tools/q8_multi.py 640 480 --ks 8,16 --wq $wq; done
  echo "== wq 1,1 opts 1,1,1"; timeout 400 python tools/q8_multi.py 640 480 --ks 8,16 --wq 1,1 --opts 1,1,1
  echo "== wq 1,1 opts 0,0,0"; timeout 400 python tools/q8_multi.py 640 480 --ks 8,16 --wq 1,1 --opts 0,0,0
  echo "== 720p wq 1,1"; timeout 400 python tools/q8_multi.py 1280 720 --ks 16 --wq 1,1 ) > gpurun_out/r05f_q8_wclip.txt 2>&1
cat gpurun_out/r05f_q8_wclip.txt
