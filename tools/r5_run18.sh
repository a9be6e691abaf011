#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp

timeout 1200 bash tools/profile_round.sh r05_register_int8_720p --dtype int8 --width 1280 --height 720 2>&1 | tail -16 | cut -c1-200
