"""CPU simulation (PyTorch fp32, tools/fp8_sim.py's bit-level emulation of the INT8 trunk) of CROSS-SCENE calibration: scales and
bias / token correction solved on hypotheses of K calibration scenes, error measured on hypotheses of held-out scenes.  Splits the
error by source: WQ=0 leaves the weights exact, AQ=0 the activations -- which of the two carries the scene-dependent common mode?
   MODE=int8c SWEEPS=2 [WQ=0|AQ=0|DITHER=1|WBITS=10] python tools/q8_sim_cross.py [refiner|scorer] [K] [hyps per scene]"""
import os, sys, time
os.environ.setdefault("MODE", "int8c")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fp8_sim as S
from foundationpose_cpp_amd import synthetic as syn, weights as W
from oracle import fp_oracle as fo, nets_torch as NT

kind = sys.argv[1] if len(sys.argv) > 1 else "refiner"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
PER = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cal = W.load_calibration(os.path.join(S.ROOT, "tests/golden/disc_calib_seed9.npz"))
mesh = syn.make_mesh(); om = fo.OracleMesh(mesh)
st = W.make_synthetic_state(kind, 9, cal)
net = NT.build(kind, st); folded = W.fold_batchnorm(st)
ratio = 1.2 if kind == "refiner" else 1.1
rng = np.random.default_rng(5)


def crops(scene, n, off):
    poses = fo.get_hyp_poses(scene.depth, scene.mask, scene.K)
    p = poses[np.arange(off, 252, 252 // n)[:n]].copy()
    if kind == "scorer":
        p[:, 12:15] += rng.normal(0, 0.003, (len(p), 3)).astype(np.float32)
    a = fo.render(om, p, scene.K, scene.depth.shape, ratio); b = fo.crop(scene.rgb, scene.depth, scene.K, p, ratio, mesh.diameter)
    return torch.from_numpy(a), torch.from_numpy(b)


cs = syn.calibration_scenes(mesh, K)
Ac, Bc = (torch.cat(t) for t in zip(*[crops(s, PER, 3 + i) for i, s in enumerate(cs)]))
tests = {"held0": syn.heldout_scenes(mesh)[0], "held1": syn.heldout_scenes(mesh)[1], "cal0(same-scene, other hyps)": cs[0]}
T = {k: crops(s, 24, 0) for k, s in tests.items()}
G = {'128': range(0, 4), '256': range(4, 8), 'b2': range(8, 9), '512': range(9, 13)}
sel = os.environ.get('LGROUPS', '128,256,b2,512').split(',')
ALL = [1 if any(i in G[g] for g in sel) else 0 for i in range(13)]
with torch.no_grad():
    exact = S.Trunk(folded, [0] * 13, True, None, h16=False)
    amax = {}
    fc = exact.forward(Ac, Bc, amax)
    exact.record = {}; exact.forward(Ac, Bc); rec_exact = exact.record; exact.record = None
    refs = {k: S.heads(net, kind, exact.forward(*v)).numpy() for k, v in T.items()}
    t = S.Trunk(folded, ALL, True, amax)
    if S.EFR:   # per-scene channel means of every activation that feeds a quantised layer (exact network, calibration hypotheses)
        means = {}
        for j in range(K):
            am = {}
            exact.means_out = am
            exact.forward(Ac[j * PER:(j + 1) * PER], Bc[j * PER:(j + 1) * PER])
            for a, v in am.items(): means.setdefault(a, []).append(v)
        exact.means_out = None
        t.scene_means = {a: torch.stack(v) for a, v in means.items()}
    t0 = time.time()
    for it in range(S.SWEEPS):
        for i, name in enumerate(S.LAYERS):
            if not ALL[i]: continue
            t.record = {}
            t.forward(Ac, Bc)
            t.bias_fix[name] = t.bias_fix.get(name, 0) + (rec_exact[name] - t.record[name])
        t.record = None
    if int(os.environ.get("TOK", "1")):
        t.tokfix = fc.mean(dim=(0, 2, 3)) - t.forward(Ac, Bc).mean(dim=(0, 2, 3))
    print(f"{kind} LGROUPS={sel} MODE={S.MODE} WQ={S.WQ} AQ={S.AQ} DITHER={S.DITHER} WBITS={S.WBITS} sweeps={S.SWEEPS} K={K}x{PER} hyps ({time.time() - t0:.0f} s)", flush=True)
    for k, v in T.items():
        out = S.heads(net, kind, t.forward(*v)).numpy()
        ref = refs[k]
        e = out - ref; de = S.dm(out) - S.dm(ref); sp = ref.std(0)
        line = f"  {k:32s} common-mode/spread {np.abs(e.mean(0) / sp).max():5.2f}  de-meaned rms/spread {(np.sqrt((de ** 2).mean(0)) / sp).max() * 100:5.1f}%"
        if kind == "refiner":
            dt = np.linalg.norm(e[:, :3], axis=1) * mesh.diameter / 2 * 1e3
            cm = np.linalg.norm(e[:, :3].mean(0)) * mesh.diameter / 2 * 1e3
            line += f"  pose err mm p95 {np.percentile(dt, 95):.2f} (common-mode {cm:.2f})"
        print(line, flush=True)
