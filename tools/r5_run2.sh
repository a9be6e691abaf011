#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_native_gpu.py -x -q 2>&1 | tail -5
timeout 600 python tools/ab_halo_wreg.py 252 > gpurun_out/r05b_ab_halo_wreg.txt 2>&1; cat gpurun_out/r05b_ab_halo_wreg.txt
( timeout 300 python tools/q8_multi.py 640 480 --ks 1,8 --amax --opts 2,1,1
  for o in 0,0,0 2,0,0 2,1,0 1,1,1; do echo "== opts $o"; timeout 300 python tools/q8_multi.py 640 480 --ks 1,8 --opts $o; done
  echo "== headroom 2.0"; timeout 300 python tools/q8_multi.py 640 480 --ks 1,8 --headroom 2.0
  echo "== headroom 2.0, no corrections"; timeout 300 python tools/q8_multi.py 640 480 --ks 1,8 --headroom 2.0 --opts 0,0,0 ) > gpurun_out/r05b_q8_opts.txt 2>&1
cat gpurun_out/r05b_q8_opts.txt
