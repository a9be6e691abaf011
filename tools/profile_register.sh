#!/bin/bash
# Round profile of the bench command: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in SEPARATE counter-only passes
# (gpurun refuses --pmc combined with trace domains).  usage: tools/profile_register.sh <tag>   -> profiles/<tag>_*
set -u
TAG=${1:-r01c}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-mfma-peak"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
cd $ROOT
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
cp "$S" gpurun_out/${TAG}_register_n252_kernel_stats.csv
python tools/summarize_pmc.py "$F" "$W" gpurun_out/${TAG}_register_n252_pmc_hbm.json
head -12 gpurun_out/${TAG}_register_n252_kernel_stats.csv
