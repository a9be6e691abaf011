"""Per-kernel time inside ONE replay of the Track hipGraph, from the rocprofv3 kernel trace tools/profile_track.sh leaves behind
(gpurun_out/prof_track/p_kernel_trace.csv): each kernel is charged the time from the previous kernel's end to its own end."""
import csv, os, sys
SEQ = "--seq" in sys.argv  # also print the kernels of the replay in order
sys.argv = [a for a in sys.argv if a != "--seq"]
f = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out/prof_track/p_kernel_trace.csv")
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
its, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    # one Track replay: from the (fused) vertex kernel to the kernel that updates the pose
    if "vertex_crop_kernel" in n and cur is None: cur = [r]
    elif cur is not None:
        cur.append(r)
        if "small_linear2_pose_kernel" in n or "token_mean_pose_kernel" in n or "enc_heads_kernel" in n: its.append(cur); cur = None
it = its[len(its) // 2]
t0 = prev = int(it[0]["Start_Timestamp"])
agg = {}
for r in it:
    e = int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0][:52]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - prev) / 1e3
    if SEQ: print(f"  {(e - prev) / 1e3:6.1f} us (own {(e - int(r['Start_Timestamp'])) / 1e3:5.1f})  grid {r.get('Grid_Size_X', '?'):>7s} wg {r.get('Workgroup_Size_X', '?'):>4s}  {k}")
    prev = e
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"{k:54s} x{n:2d} {t:7.1f} us")
print("kernels", len(it), "span", (prev - t0) / 1e3, "us")
