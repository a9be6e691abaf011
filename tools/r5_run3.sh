#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ab_halo_wreg.py 252 > gpurun_out/r05c_ab_halo_wreg.txt 2>&1; cat gpurun_out/r05c_ab_halo_wreg.txt
timeout 600 python tools/ab_pipeline.py fpt_set_halo_wreg 0 1 0 1 > gpurun_out/r05c_ab_pipeline_wreg.txt 2>&1; cat gpurun_out/r05c_ab_pipeline_wreg.txt
timeout 600 python tools/profile_register_stages.py > gpurun_out/r05c_stages.txt 2>&1; tail -60 gpurun_out/r05c_stages.txt
