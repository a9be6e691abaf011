#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( for tau in 0.2 0.3 0.45; do for lam in 0.01 0.05 0.3; do echo "== tau $tau lam $lam"; FP_Q8_TAU=$tau FP_Q8_LAM=$lam HELD=8 timeout 600 python tools/q8_multi.py 640 480 --ks 16 --wq 1,2; FP_Q8_TAU=$tau FP_Q8_LAM=$lam HELD=8 timeout 600 python tools/q8_multi.py 1280 720 --ks 16 --wq 1,2; done; done ) > gpurun_out/r05i_q8_grid.txt 2>&1
python - <<'PY'
import re
for line in open('gpurun_out/r05i_q8_grid.txt'):
    if line.startswith('=='): print(line.strip(), end='  ')
    elif line.startswith('int8'):
        sh=[float(x) for x in re.findall(r'(\d+\.\d)% p95', line)]; cm=[float(x) for x in re.findall(r'cm (\d+\.\d+)', line)]
        print(line[:18], 'share', ' '.join(f'{x:5.1f}' for x in sh), '| min %.1f mean %.1f | cm mean %.2f max %.2f' % (min(sh), sum(sh)/len(sh), sum(cm)/len(cm), max(cm)))
PY
