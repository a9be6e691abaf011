#!/bin/bash
# Race hunt: two models on two threads WITHOUT the process-wide lock, under different runtime settings.
mkdir -p gpurun_out/race
run() { tag=$1; shift; echo "=== $tag"; ( env FP_DISABLE_GPU_LOCK=1 "$@" timeout 300 python tools/dbg_concurrent.py 0 ${ITERS:-40} > gpurun_out/race/$tag.log 2>&1; echo "rc=$?" >> gpurun_out/race/$tag.log ); grep -c "digests differ" gpurun_out/race/$tag.log; tail -4 gpurun_out/race/$tag.log; }
run base
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run sdma0 HSA_ENABLE_SDMA=0
run hwq1 GPU_MAX_HW_QUEUES=1
run serialize AMD_SERIALIZE_KERNEL=3
