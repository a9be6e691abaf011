#!/bin/bash
# per-dispatch durations of the kernels matching a name filter, by grid size, from a rocprofv3 kernel trace of a bench.py command
# usage: tools/trace_kernel.sh <name filter> [bench.py arguments]
set -u
PAT=$1; shift
ROOT=$(pwd); export TMPDIR=/tmp
OUT=/tmp/trace_$$; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-mfma-peak --no-extras "$@" > $OUT/log 2>&1
cd $ROOT
T=$(find $OUT -name '*kernel_trace.csv' | head -1)
python - "$T" "$PAT" <<'EOF'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"][:60], r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    v = sorted(v)
    print(k, "n", len(v), "median us %.1f" % v[len(v) // 2], "min %.1f" % v[0], "max %.1f" % v[-1])
EOF
rm -rf $OUT
