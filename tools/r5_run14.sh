#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -30 > gpurun_out/r05m_gputests.txt; tail -12 gpurun_out/r05m_gputests.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>gpurun_out/r05m_bench_err.txt | tail -1 > gpurun_out/r05m_bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05m_bench_default.json'))
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('int8_720p','int8_720p_untextured','track','track_int8','track_bf16','n1008','host_frame'):
    e=d.get(k,{})
    print(k, e.get('value'), e.get('ms_per_step', e.get('ms_per_frame')), json.dumps(e.get('accuracy',{}).get('pose_delta_vs_f16',{})), e.get('accuracy',{}).get('common_mode_mm'), e.get('accuracy',{}).get('winner_rank_teacher_forced'))
print(json.dumps(d.get('int8_720p',{}).get('stage_ms')))
PY
