#!/bin/bash
# INT8 held-out accuracy: per-channel range = max over the calibration frames + kappa * sd of the per-frame maxima
mkdir -p gpurun_out
for k in 0 1 2 3 5; do
  HELD=8 python tools/q8_multi.py --ks 16 --kappa $k 2>&1 | grep "K=16" | sed "s/^/kappa $k: /" >> gpurun_out/q8_kappa.txt
done
HELD=8 python tools/q8_multi.py --ks 16 --kappa 2 --headroom 1.3125 2>&1 | grep "K=16" | sed "s/^/kappa 2 x1.05: /" >> gpurun_out/q8_kappa.txt
python - <<'PY'
import re
for l in open('gpurun_out/q8_kappa.txt'):
    tag=l.split(': int8')[0]
    sh=[float(x) for x in re.findall(r'(\d+\.\d)% p95', l)]
    cm=[float(x) for x in re.findall(r'cm (\d+\.\d+)', l)]
    p95=[float(x) for x in re.findall(r'p95 (\d+\.\d+) max', l)]
    print(f"{tag:16s} share mean {sum(sh)/len(sh):5.1f} min {min(sh):5.1f} | cm mean {sum(cm)/len(cm):.2f} max {max(cm):.2f} | p95 mean {sum(p95)/len(p95):.2f} :: "+" ".join(f"{x:.0f}" for x in sh))
PY
