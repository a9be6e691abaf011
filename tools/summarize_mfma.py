#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from one rocprofv3 --pmc pass (tools/profile_round.sh).

  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.
              SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the chip (16 cycles per v_mfma_f32_16x16x32_f16, 32 per
              v_mfma_f32_16x16x128_f8f6f4: check mfma_busy_cycles_per_launch / mfma_insts_per_launch), GRBM_GUI_ACTIVE over
              the 8 XCDs; the ratio is the share of the matrix pipes' cycles spent executing MFMAs AT THE CLOCK THE KERNEL
              RAN AT (clock_ghz, from the kernel-trace pass's duration) -- multiply by clock_ghz / 2.4 to compare with a
              fraction of the 2.4 GHz datasheet peak.
  mops      = SQ_INSTS_VALU_MFMA_MOPS_{F16,BF16,F8} per launch (512 FLOP per MOP for the 2-byte types; the counter also
              lets one check that an FP8 kernel really issues FP8 MFMAs)
The average duration from the kernel-trace pass of the same command is joined in for reference.
usage: summarize_mfma.py <counter_collection.csv> <kernel_stats.csv> <out.json>
"""
import collections
import csv
import json
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    dur = {}
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            dur[r["Name"].split("(")[0].strip()] = (int(r["Calls"]), float(r["AverageNs"]))
    out = {}
    for k, c in agg.items():
        n = max(len(launches[k]), 1)
        busy, mf = c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / n / 8.0
        ns = dur.get(k, (0, None))[1]
        out[k] = dict(launches=n, mfma_util=(mf / n / (1024.0 * gui) if gui else None),
                      clock_ghz=(gui / ns if ns else None),
                      mfma_busy_cycles_per_launch=mf / n, sq_busy_cycles_per_launch=busy / n,
                      wave_cycles_per_launch=c.get("SQ_WAVE_CYCLES", 0.0) / n, mfma_insts_per_launch=c.get("SQ_INSTS_MFMA", 0.0) / n,
                      mops_f16_per_launch=c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) / n,
                      mops_bf16_per_launch=c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / n,
                      mops_f8_per_launch=c.get("SQ_INSTS_VALU_MFMA_MOPS_F8", 0.0) / n,
                      gui_active_per_launch=c.get("GRBM_GUI_ACTIVE", 0.0) / n,
                      avg_ns_kernel_trace=dur.get(k, (0, None))[1])
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))[:10]:
        mb = v["mfma_util"]
        print(f"{k[:56]:56s} launches={v['launches']:4d} mfma_util={mb if mb is None else round(mb, 3)} clock={v['clock_ghz'] and round(v['clock_ghz'], 2)} GHz "
              f"mops f16/bf16/f8 per launch = {v['mops_f16_per_launch']:.3g} / {v['mops_bf16_per_launch']:.3g} / {v['mops_f8_per_launch']:.3g}")


if __name__ == "__main__":
    main()
