"""How the rasteriser's shading pass scales with the TEXTURE: Register (N = 252) and Track with textures of 512^2 (the synthetic
default) to 4096^2 texels (YCB texture maps are 2048^2 - 4096^2).   python tools/texture_size_sweep.py [size ...]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, synthetic as syn, weights as W
d = tempfile.mkdtemp(); rp, sp = os.path.join(d, "r.fpw"), os.path.join(d, "s.fpw")
W.pack_synthetic("refiner", rp); W.pack_synthetic("scorer", sp)
sub = int(os.environ.get("SUBDIV", 5))
for size in [int(a) for a in sys.argv[1:]] or [512, 2048, 4096]:
    mesh = syn.make_mesh(subdiv=sub)
    mesh.texture = syn.make_texture(size=size)
    scene = syn.make_scene(mesh)
    m = FoundationPose(mesh, scene.K, rp, sp)
    hyp = syn.perturb_pose(scene.gt_pose)
    out = [f"texture {size}^2, {len(mesh.faces)} triangles:"]
    for name, call, n in (("Track", lambda: m.Track(scene.rgb, scene.depth, hyp, mesh.name), 20),
                          ("Register", lambda: m.Register(scene.rgb, scene.depth, scene.mask, mesh.name), 4)):
        for _ in range(2): call()
        m.profile(True); m.profile_reset()
        for _ in range(n): call()
        r = m.profile_report(); m.profile(False)
        out.append(f"{name} raster_shade {r['raster_shade']['ms'] * 1e3 / n:.1f} us")
    print("  ".join(out))
    m.close()
