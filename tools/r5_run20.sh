#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --no-cpu-baseline 2>gpurun_out/r05o_bench_err.txt | tail -1 > gpurun_out/r05o_bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05o_bench_default.json'))
print('bench', d['value'], d['ms_per_step'])
for k in ('int8_720p','int8_720p_untextured','track_int8'):
    e=d.get(k,{}); print(k, e.get('value'), e.get('ms_per_step', e.get('ms_per_frame')), e.get('accuracy',{}).get('pose_delta_vs_f16',{}).get('frac_within_1mm_1deg'), e.get('accuracy',{}).get('common_mode_mm'))
print(json.dumps(d['int8_720p']['stage_ms']))
PY
timeout 600 python -m pytest tests/test_precision_gpu.py -q -x -k "heldout or session or small_batches" 2>&1 | grep "passed\|failed" | tail -2
