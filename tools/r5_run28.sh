#!/bin/bash
# INT8 held-out accuracy against the number of calibration frames (headroom 1.25, the product build)
mkdir -p gpurun_out
HELD=8 python tools/q8_multi.py --ks 8,16,24,32,48,64 2>&1 | grep "K=" > gpurun_out/q8_ks.txt
HELD=8 python tools/q8_multi.py --ks 32,48 --untextured 2>&1 | grep "K=" | sed "s/^/untextured /" >> gpurun_out/q8_ks.txt
HELD=8 python tools/q8_multi.py 1280 720 --ks 16,32 2>&1 | grep "K=" >> gpurun_out/q8_ks.txt
python - <<'PY'
import re
for l in open('gpurun_out/q8_ks.txt'):
    tag=l.split(' (')[0]
    sh=[float(x) for x in re.findall(r'(\d+\.\d)% p95', l)]
    cm=[float(x) for x in re.findall(r'cm (\d+\.\d+)', l)]
    p95=[float(x) for x in re.findall(r'p95 (\d+\.\d+) max', l)]
    print(f"{tag:34s} share mean {sum(sh)/len(sh):5.1f} min {min(sh):5.1f} | cm mean {sum(cm)/len(cm):.2f} max {max(cm):.2f} | p95 mean {sum(p95)/len(p95):.2f} :: "+" ".join(f"{x:.0f}" for x in sh))
PY
