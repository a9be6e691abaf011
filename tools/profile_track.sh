#!/bin/bash
# rocprofv3 kernel-trace of Track (hipGraph replay): per-kernel durations inside the graph
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_track; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $ROOT/bench.py --track --steps 200 --warmup 20 --no-cpu-baseline --no-extras $* > $OUT/log.txt 2>&1
cd $ROOT
tail -1 $OUT/log.txt | cut -c1-300
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_track/**/p_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n_track = 220
print(f"kernel time per Track: {tot/n_track/1e3:.1f} us over {sum(int(r['Calls']) for r in rows)/n_track:.1f} kernels")
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls/track {int(r['Calls'])/n_track:5.1f}  avg {float(r['AverageNs'])/1e3:7.2f} us  total/track {float(r['TotalDurationNs'])/n_track/1e3:7.1f} us")
PY
