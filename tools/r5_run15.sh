#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_precision_gpu.py tests/test_nn_gpu.py tests/test_properties_gpu.py -q -x -k "heldout or small_batches or track or Track or fusions" 2>&1 | tail -5
bash tools/profile_track.sh 2>&1 | tail -3
python tools/track_timeline.py > gpurun_out/r05n_track_timeline.txt 2>&1; tail -14 gpurun_out/r05n_track_timeline.txt
timeout 900 python bench.py --steps 10 2>gpurun_out/r05n_bench_err.txt | tail -1 > gpurun_out/r05n_bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05n_bench_default.json'))
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('int8_720p','int8_720p_untextured','track','track_int8','track_bf16','n1008','host_frame'):
    e=d.get(k,{})
    print(k, e.get('value'), e.get('ms_per_step', e.get('ms_per_frame')), e.get('host_frame_value'), json.dumps(e.get('accuracy',{}).get('pose_delta_vs_f16',{}).get('frac_within_1mm_1deg')), e.get('accuracy',{}).get('common_mode_mm'))
PY
