#!/bin/bash
# Round evidence for one bench command: rocprofv3 kernel-trace stats, then counter-only passes (gpurun refuses --pmc combined
# with trace domains): FETCH_SIZE, WRITE_SIZE, and the matrix-pipe counters (SURVEY.md section 8d "Evidence").
# usage: tools/profile_round.sh <tag> [bench.py arguments]   ->   gpurun_out/<tag>_*  (copy what you keep into profiles/)
set -u
TAG=${1:-r02}
shift
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-mfma-peak --no-extras $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F8 GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/mfma -o p -- $CMD > $OUT/mfma.log 2>&1
cd $ROOT
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
M=$(find $OUT/mfma -name '*counter_collection.csv' | head -1)
cp "$S" gpurun_out/${TAG}_kernel_stats.csv
python tools/summarize_pmc.py "$F" "$W" gpurun_out/${TAG}_pmc_hbm.json
python tools/summarize_mfma.py "$M" "$S" gpurun_out/${TAG}_pmc_mfma.json
head -14 gpurun_out/${TAG}_kernel_stats.csv
rm -rf $OUT   # (the raw traces of a run that calibrates are > 64 MiB: gpurun would not copy anything back)
