#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( for wq in 0,0,0 0,0,1 1,2,0 1,2,1; do echo "== wq $wq opts 0,0,0"; HELD=4 timeout 600 python tools/q8_multi.py 640 480 --ks 16 --wq $wq --opts 0,0,0; done ) > gpurun_out/r05k_q8_diag.txt 2>&1
python - <<'PY'
import re
for line in open('gpurun_out/r05k_q8_diag.txt'):
    if line.startswith('=='): print(line.strip())
    elif line.startswith('int8'):
        sh=[float(x) for x in re.findall(r'(\d+\.\d)% p95', line)]; cm=[float(x) for x in re.findall(r'cm (\d+\.\d+)', line)]; p95=[float(x) for x in re.findall(r'p95 (\d+\.\d+) max', line)]
        print(line[:28], 'cm', ' '.join(f'{x:5.2f}' for x in cm), '| p95', ' '.join(f'{x:5.2f}' for x in p95))
    else: print(line.strip()[:300])
PY
