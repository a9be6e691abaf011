"""How the geometry stages scale with the mesh: Track and Register (N = 252) with icospheres of 5 k / 20 k / 82 k triangles
(the synthetic default is 5 120; YCB `textured_simple` meshes have ~16 k).   python tools/mesh_size_sweep.py [subdiv ...]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
for sub in [int(a) for a in sys.argv[1:]] or [4, 5, 6]:
    mesh = syn.make_mesh(subdiv=sub); scene = syn.make_scene(mesh)
    m = FoundationPose(mesh, scene.K, rp, sp)
    hyp = syn.perturb_pose(scene.gt_pose)
    out = [f"{len(mesh.faces)} triangles:"]
    for name, call, n in (("Track", lambda: m.Track(scene.rgb, scene.depth, hyp, mesh.name), 20),
                          ("Register", lambda: m.Register(scene.rgb, scene.depth, scene.mask, mesh.name), 4)):
        for _ in range(2): call()
        m.profile(True); m.profile_reset()
        for _ in range(n): call()
        r = m.profile_report(); m.profile(False)
        geo = {k: v["ms"] * 1e3 / n for k, v in r.items() if k in ("raster_shade", "tri_rows", "vertex", "vertex_crop", "crop_warp")}
        for _ in range(3): call()
        t0 = time.perf_counter()
        for _ in range(n): call()
        wall = (time.perf_counter() - t0) / n * 1e6
        out.append(f"{name} {wall:.0f} us/call (" + ", ".join(f"{k} {v:.1f}" for k, v in sorted(geo.items())) + " us)")
    print("  ".join(out))
    m.close()
