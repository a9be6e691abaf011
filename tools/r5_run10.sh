#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( for wq in 1,2,1 1,2,0 0,0,1; do echo "== wq $wq"; HELD=8 timeout 600 python tools/q8_multi.py 640 480 --ks 1,8,16 --wq $wq; HELD=8 timeout 600 python tools/q8_multi.py 1280 720 --ks 16 --wq $wq; done ) > gpurun_out/r05j_q8_imgbias.txt 2>&1
python - <<'PY'
import re
for line in open('gpurun_out/r05j_q8_imgbias.txt'):
    if line.startswith('=='): print(line.strip())
    elif line.startswith('int8'):
        sh=[float(x) for x in re.findall(r'(\d+\.\d)% p95', line)]; cm=[float(x) for x in re.findall(r'cm (\d+\.\d+)', line)]; p95=[float(x) for x in re.findall(r'p95 (\d+\.\d+) max', line)]
        print(line[:28], 'share', ' '.join(f'{x:5.1f}' for x in sh), '| min %.1f mean %.1f | cm mean %.2f max %.2f | p95 mean %.2f' % (min(sh), sum(sh)/len(sh), sum(cm)/len(cm), max(cm), sum(p95)/len(p95)))
    else: print(line.strip()[:300])
PY
timeout 900 python -m pytest tests/test_precision_gpu.py -x -q 2>&1 | tail -8
