#!/bin/bash
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_f16.err | tail -1 > gpurun_out/bench_f16.json
python bench.py --steps 20 --warmup 5 --dtype fp8 --width 1280 --height 720 --no-cpu-baseline 2>gpurun_out/bench_fp8.err | tail -1 > gpurun_out/bench_fp8_720p.json
python bench.py --steps 20 --warmup 5 --dtype f16 --width 1280 --height 720 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_f16_720p.json
python bench.py --steps 20 --warmup 5 --dtype fp8 --width 1280 --height 720 --untextured --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_fp8_720p_untextured.json
python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_bf16.json
for f in f16 fp8_720p f16_720p fp8_720p_untextured bf16; do python - <<PY
import json
d=json.load(open("gpurun_out/bench_$f.json"))
r=d["roofline"]
print("$f", d["value"], d["ms_per_step"], "dom", r["kernel"], r["achieved"], r["frac"], "family", r["conv_family"]["frac"], r["conv_family"]["ms_per_step"], "host", d.get("host_frame",{}).get("value"), "track", d.get("track",{}).get("value"), d.get("track",{}).get("host_frame_value"))
print("   ", r["conv_family"]["kernels_ms"])
print("   ", {k:v for k,v in list(d["stage_ms"].items())[:12]})
PY
done
tail -3 gpurun_out/bench_f16.err gpurun_out/bench_fp8.err
