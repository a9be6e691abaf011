#!/bin/bash
# INT8 held-out accuracy: K = 32 calibration frames with less activation headroom
mkdir -p gpurun_out
for h in 1.0 1.1 1.25; do
  HELD=8 python tools/q8_multi.py --ks 32 --headroom $h 2>&1 | grep "K=32" | sed "s/^/K32 headroom $h: /" >> gpurun_out/q8_k32.txt
done
HELD=8 python tools/q8_multi.py --ks 16 --headroom 1.1 2>&1 | grep "K=16" | sed "s/^/K16 headroom 1.1: /" >> gpurun_out/q8_k32.txt
python - <<'PY'
import re
for l in open('gpurun_out/q8_k32.txt'):
    tag=l.split(': int8')[0]
    sh=[float(x) for x in re.findall(r'(\d+\.\d)% p95', l)]
    cm=[float(x) for x in re.findall(r'cm (\d+\.\d+)', l)]
    p95=[float(x) for x in re.findall(r'p95 (\d+\.\d+) max', l)]
    print(f"{tag:20s} share mean {sum(sh)/len(sh):5.1f} min {min(sh):5.1f} | cm mean {sum(cm)/len(cm):.2f} max {max(cm):.2f} | p95 mean {sum(p95)/len(p95):.2f} :: "+" ".join(f"{x:.0f}" for x in sh))
PY
