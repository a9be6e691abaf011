export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_properties_gpu.py -m gpu -x -q -k "fusions or row_ranges" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_golden_gpu.py tests/test_demo_gpu.py -m gpu -x -q -k "track or Track or golden or demo" 2>&1 | tail -5
for i in 1 2; do timeout 300 python tools/ab_track.py fpt_set_fuse_pose 1 2 0 2>&1 | grep "us per"; timeout 300 python tools/ab_track.py fpt_set_vertex_crop 1 2 0 2>&1 | grep "us per"; done
bash tools/profile_track.sh 2>&1 | tail -30
python tools/track_timeline.py > gpurun_out/r04h_track_timeline.txt 2>&1; cat gpurun_out/r04h_track_timeline.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04h_bench_default.json
python -c "import json; d=json.load(open('gpurun_out/r04h_bench_default.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac']); print({k:(d[k]['value'],d[k].get('ms_per_frame')) for k in ('track','track_int8','track_bf16')}); print(d['track'])"
