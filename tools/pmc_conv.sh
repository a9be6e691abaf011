#!/bin/bash
# PMC pass over the conv micro-benchmark (counters only; no trace domains, as gpurun requires)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o p -- python tools/bench_conv.py --variant ${1:-1} --iters 2 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_conv/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("(")[0][-30:], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT"): cnt[(key, r["Counter_Name"])] += 1
for key, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = max(cnt[(key, "SQ_WAVE_CYCLES")], 1)
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(key, "launches", n)
    for c in sorted(d):
        print(f"    {c:28s} {d[c]/n:14.4e}  ({100*d[c]/wc:6.1f}% of WAVE_CYCLES)")
PY
