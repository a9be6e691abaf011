"""CPU simulation of the FP8 (OCP e4m3) trunk under the DISCRIMINATING weight set: which quantisation design reaches the parity
bars of BASELINE configs[4]?  (round 4; sets the design of the FP8 path -- DESIGN.md section 4.4.)

Emulates, in PyTorch fp32 on the CPU, exactly what the HIP kernels do to the 13 FP8 trunk convolutions: per-output-channel weight
scales (amax/448, RNE, saturating), static per-tensor activation scales (amax of a calibration batch / 224), products accumulated
in fp32.  Variants:
  stream=fp8   every trunk tensor stored in e4m3 (round-3 design: the skip path is re-quantised by every residual block)
  stream=f16   block outputs kept in f16 for the skip path, e4m3 copy only as the next conv's operand
  bc=1         per-channel bias correction (Nagel et al. 2019): every FP8 layer's bias absorbs the mean output error measured on
               the calibration batch, layer by layer
  mask         which of the 13 layers run in FP8 (the others f16 = exact here)
Reports de-meaned error / between-hypothesis spread, correlation of the deltas and refined-pose distances vs the fp32 network.

   python tools/fp8_sim.py [refiner|scorer] ...
"""
import os, sys, time, itertools
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from foundationpose_cpp_amd import synthetic as syn, weights as W
from oracle import fp_oracle as fo, nets_torch as NT

torch.set_num_threads(os.cpu_count())
LAYERS = ["encodeA.2.conv1", "encodeA.2.conv2", "encodeA.3.conv1", "encodeA.3.conv2", "encodeAB.0.conv1", "encodeAB.0.conv2",
          "encodeAB.1.conv1", "encodeAB.1.conv2", "encodeAB.2", "encodeAB.3.conv1", "encodeAB.3.conv2", "encodeAB.4.conv1", "encodeAB.4.conv2"]


MODE = os.environ.get("MODE", "fp8")   # fp8 | int8 (per-tensor u8 activations) | int8c (per-input-channel u8 activations)
AMARGIN = float(os.environ.get("AMARGIN", "1.25"))
POSTBC = int(os.environ.get("POSTBC", "0")); POSTGAIN = float(os.environ.get("POSTGAIN", "1.0"))
CLIPK = float(os.environ.get("CLIPK", "0")); FLOOR = float(os.environ.get("FLOOR", "0"))
SWEEPS = int(os.environ.get("SWEEPS", "1")); JACOBI = int(os.environ.get("JACOBI", "0"))
WQ = int(os.environ.get("WQ", "1")); AQ = int(os.environ.get("AQ", "1"))   # [r5] int8c only: quantise the weights / the activations (0 = leave exact: which of the two carries the scene-dependent bias)
DITHER = int(os.environ.get("DITHER", "0"))   # [r5] int8 activations: floor(x / s + u), u ~ U[0,1) from a fixed per-(pixel, channel) pattern (unbiased rounding)
EFR = int(os.environ.get("EFR", "0"))         # [r5] error-feedback rounding of the folded weights against the per-scene channel means of the layer's input
EFR_TAU = float(os.environ.get("EFR_TAU", "0.5")); EFR_LAM = float(os.environ.get("EFR_LAM", "0.0"))
WBITS = int(os.environ.get("WBITS", "8"))     # [r5] bits of the (folded) weights: 8 = the kernels' int8 rows


def q8(x, scale):
    return (x / scale).clamp(-448, 448).to(torch.float8_e4m3fn).float() * scale


def qw(w):
    if MODE == "fp8":
        return q8(w, w.abs().amax(dim=(1, 2, 3), keepdim=True) / 448.0)
    sc = w.abs().amax(dim=(1, 2, 3), keepdim=True) / 127.0
    return (w / sc).round().clamp(-127, 127) * sc


def qa(x, amax_t, amax_c):
    """activation quantiser: x >= 0 (post-ReLU)"""
    if MODE == "fp8":
        return q8(x, amax_t / 224.0)
    if MODE == "int8":
        sc = amax_t * AMARGIN / 255.0
    else:
        sc = (amax_c * AMARGIN / 255.0).clamp_min(1e-8).view(1, -1, 1, 1)
    if not AQ: return x
    if DITHER:
        g = torch.Generator().manual_seed(x.shape[1] * 7919 + x.shape[2])
        u = torch.rand(x.shape[1:], generator=g)
        return (x / sc + u).floor().clamp(0, 255) * sc
    return (x / sc).round().clamp(0, 255) * sc


def efr_round(v, m, qmax):
    """v [Cout,Cin,kh,kw] = folded weights in units of the row's step; m [J,Cin] = mean INTEGER activation of every input channel in
    each of J calibration scenes.  Round every weight up or down (only where its fraction is within EFR_TAU of .5 is the choice free)
    so that, per output row, sum_k (q_k - v_k) m_jc(k) stays near zero for every scene j: the part of the weight error that shifts the
    layer's mean output -- and shifts it differently in every scene -- cancels instead of accumulating like a random walk."""
    Cout, Cin, kh, kw = v.shape
    base = v.floor(); frac = v - base
    near = base + (frac >= 0.5).float()
    q = near.clone()
    r = torch.zeros(Cout, m.shape[0], dtype=torch.float64)
    free = (frac - 0.5).abs() < EFR_TAU
    md = m.double()
    for c in range(Cin):
        mc = md[:, c]                                     # [J]
        for ky in range(kh):
            for kx in range(kw):
                f = frac[:, c, ky, kx].double()
                e_dn, e_up = -f, 1.0 - f                  # error (q - v) of rounding down / up
                c_dn = ((r + e_dn[:, None] * mc[None, :]) ** 2).sum(1) + EFR_LAM * e_dn ** 2
                c_up = ((r + e_up[:, None] * mc[None, :]) ** 2).sum(1) + EFR_LAM * e_up ** 2
                up = torch.where(free[:, c, ky, kx], c_up < c_dn, f >= 0.5)
                e = torch.where(up, e_up, e_dn)
                r += e[:, None] * mc[None, :]
                q[:, c, ky, kx] = base[:, c, ky, kx] + up.float()
    return q.clamp(-qmax, qmax)


HALF = os.environ.get("HALF", "f16")   # storage type of the non-8-bit tensors: f16 | bf16 (the bf16 precision's arithmetic limit)


def qh(x):
    return x.bfloat16().float() if HALF == "bf16" else x.half().float()


class Trunk:
    """functional trunk on folded weights; fp8[i] says whether LAYERS[i] runs on e4m3 operands"""
    def __init__(self, folded, fp8, stream_f16, act_amax=None, h16=True):
        self.w = {k: torch.from_numpy(v) for k, v in folded.items()}
        self.fp8 = list(fp8); self.stream_f16 = stream_f16; self.amax = act_amax; self.h16 = h16
        self.bias_fix = {}
        self.tokfix = None
        self.wq = {}
        for i, name in enumerate(LAYERS):
            w = self.w[name + ".weight"]
            if self.fp8[i]:
                self.wq[name] = qw(w)
            else:
                self.wq[name] = qh(w) if h16 else w
        self.record = None  # dict name -> per-channel mean of the layer's PRE-activation output

    def conv(self, name, x, stride, i=None, act_id=None):
        w = self.wq.get(name)
        if w is None:
            w = qh(self.w[name + ".weight"]) if self.h16 else self.w[name + ".weight"]
        if MODE == "int8c" and name in LAYERS and self.fp8[LAYERS.index(name)]:
            # exact emulation: the per-input-channel activation scale s_c is folded into the weights BEFORE they are quantised
            key = ("fold", name)
            if key not in self.wq:
                sc = (self.amax[("c", act_id)] * AMARGIN / 255.0).clamp_min(1e-8).view(1, -1, 1, 1)
                wf = self.w[name + ".weight"] * sc
                qmax = float(2 ** (WBITS - 1) - 1)
                sw = wf.abs().amax(dim=(1, 2, 3), keepdim=True) / qmax
                if EFR and WQ and getattr(self, "scene_means", None) is not None:
                    q = efr_round(wf / sw, self.scene_means[act_id] / sc.view(1, -1), qmax)
                    self.wq[key] = q * sw / sc
                else:
                    self.wq[key] = ((wf / sw).round().clamp(-qmax, qmax) * sw / sc) if WQ else self.w[name + ".weight"]
            w = self.wq[key]
        b = self.w[name + ".bias"]
        if name in self.bias_fix: b = b + self.bias_fix[name]
        k = w.shape[-1]
        y = F.conv2d(x, w, b, stride, (k - 1) // 2)
        return y

    def act_in(self, x, layer_idx, act_id):
        """operand of LAYERS[layer_idx]: e4m3 with the static per-tensor scale of activation act_id, or f16"""
        if self.fp8[layer_idx]:
            return qa(x, self.amax[act_id], self.amax.get(("c", act_id)))
        return qh(x) if self.h16 else x

    def store(self, x, consumer_idx, act_id, is_stream):
        """what the producing layer leaves in memory for the SKIP path / later consumers"""
        if consumer_idx is not None and self.fp8[consumer_idx] and not (is_stream and self.stream_f16):
            return qa(x, self.amax[act_id], self.amax.get(("c", act_id)))
        return qh(x) if self.h16 else x

    def block(self, x, li, act_mid, act_out, next_idx, post=None):
        """ResnetBasicBlock: LAYERS[li], LAYERS[li+1]; x = stored stream tensor (activation id act_mid-1)"""
        n1, n2 = LAYERS[li], LAYERS[li + 1]
        h = torch.relu(self.pre(n1, self.conv(n1, self.act_in(x, li, act_mid - 1), 1, act_id=act_mid - 1)))
        h = self.store(h, li + 1, act_mid, False)
        y = self.pre(n2, self.conv(n2, self.act_in(h, li + 1, act_mid), 1, act_id=act_mid), res=x)
        y = torch.relu(y + x)
        if post is not None: y = post(y)
        return self.store(y, next_idx, act_out, True)

    def pre(self, name, y, res=None):
        if self.record is not None:
            self.record[name] = y.mean(dim=(0, 2, 3)).clone()
        return y

    def forward(self, A, B, amax_out=None):
        bs = len(A)
        x = torch.cat([A, B], 0).permute(0, 3, 1, 2)
        x = qh(x) if self.h16 else x
        x = torch.relu(self.conv("encodeA.0", x, 2)); x = qh(x) if self.h16 else x
        x = torch.relu(self.conv("encodeA.1", x, 2))
        acts = {}
        def note(i, t):
            if getattr(self, "means_out", None) is not None: self.means_out[i] = t.mean(dim=(0, 2, 3)).clone()
            if amax_out is not None:
                amax_out[i] = max(amax_out.get(i, 0.0), float(t.abs().max()))
                c = t.abs().amax(dim=(0, 2, 3))
                if CLIPK > 0:   # clip the range at mean + K sigma of the channel (never above its |max|)
                    c = torch.minimum(c, t.mean(dim=(0, 2, 3)) + CLIPK * t.std(dim=(0, 2, 3)))
                if FLOOR > 0: c = torch.maximum(c, c.max() * FLOOR)
                amax_out[("c", i)] = torch.maximum(amax_out[("c", i)], c) if ("c", i) in amax_out else c
        note(1, x)
        x = self.store(x, 0, 1, True)
        # encodeA blocks (acts 2,3 | 4,5)
        x = self._blk(x, 0, 2, 3, 2, note)
        x = self._blk(x, 2, 4, 5, 4, note, cat=bs)
        x = self._blk(x, 4, 6, 7, 6, note)
        x = self._blk(x, 6, 8, 9, 8, note)
        # b2
        y = torch.relu(self.pre(LAYERS[8], self.conv(LAYERS[8], self.act_in(x, 8, 9), 2, act_id=9)))
        if self.record is not None: self.record["post:" + LAYERS[8]] = y.mean(dim=(0, 2, 3)).clone()
        note(10, y)
        x = self.store(y, 9, 10, True)
        x = self._blk(x, 9, 11, 12, 11, note)
        x = self._blk(x, 11, 13, 14, None, note)
        if self.tokfix is not None: x = x + self.tokfix.view(1, -1, 1, 1)
        return x  # [bs,512,20,20] f16 tokens (before positional embedding)

    def _blk(self, x, li, act_mid, act_out, next_idx, note, cat=None):
        n1, n2 = LAYERS[li], LAYERS[li + 1]
        h = torch.relu(self.pre(n1, self.conv(n1, self.act_in(x, li, act_mid - 1), 1, act_id=act_mid - 1)))
        if self.record is not None: self.record["post:" + n1] = h.mean(dim=(0, 2, 3)).clone()
        note(act_mid, h)
        h = self.store(h, li + 1, act_mid, False)
        y = torch.relu(self.pre(n2, self.conv(n2, self.act_in(h, li + 1, act_mid), 1, act_id=act_mid)) + x)
        if self.record is not None: self.record["post:" + n2] = y.mean(dim=(0, 2, 3)).clone()
        if cat is not None:
            y = torch.cat((y[:cat], y[cat:]), 1).contiguous()
        note(act_out, y)
        return self.store(y, next_idx, act_out, True)


def heads(net, kind, feat):
    tok = net.pos_embed(feat.reshape(feat.shape[0], feat.shape[1], -1).permute(0, 2, 1))
    if kind == "refiner":
        return torch.cat([net.trans_head(tok).mean(1), net.rot_head(tok).mean(1)], 1)
    ab, _ = net.att(tok, tok, tok, need_weights=False)
    return net.head(ab.mean(1))[:, None]


def dm(x): return x - x.mean(0, keepdims=True)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "refiner"
    cal = W.load_calibration(os.path.join(ROOT, "tests/golden/disc_calib_seed9.npz"))
    mesh = syn.make_mesh(); scene = syn.make_scene(mesh); om = fo.OracleMesh(mesh)
    st = W.make_synthetic_state(kind, 9, cal)
    net = NT.build(kind, st); folded = W.fold_batchnorm(st)
    poses = fo.get_hyp_poses(scene.depth, scene.mask, scene.K)
    ratio = 1.2 if kind == "refiner" else 1.1
    rng = np.random.default_rng(5)
    def crops(sel):
        p = poses[sel].copy()
        if kind == "scorer":  # score-mode inputs are refined poses: jitter the translation like a refine step does
            p[:, 12:15] += rng.normal(0, 0.003, (len(p), 3)).astype(np.float32)
        a = fo.render(om, p, scene.K, scene.depth.shape, ratio); b = fo.crop(scene.rgb, scene.depth, scene.K, p, ratio, mesh.diameter)
        return torch.from_numpy(a), torch.from_numpy(b)
    n_test = int(os.environ.get("NTEST", "42")); n_cal = int(os.environ.get("NCAL", "24"))
    test_sel = np.arange(0, 252, 252 // n_test)[:n_test]; cal_sel = np.arange(3, 252, 252 // n_cal)[:n_cal]
    At, Bt = crops(test_sel); Ac, Bc = crops(cal_sel)
    t0 = time.time()
    with torch.no_grad():
        exact = Trunk(folded, [0] * 13, True, None, h16=False)
        amax = {}
        fc = exact.forward(Ac, Bc, amax)
        exact.record = {}; exact.forward(Ac, Bc); rec_exact = exact.record; exact.record = None
        ref = heads(net, kind, exact.forward(At, Bt)).numpy()
        f16 = heads(net, kind, Trunk(folded, [0] * 13, True, None, h16=True).forward(At, Bt)).numpy()
    print(f"{kind}: fp32 reference {time.time()-t0:.1f}s; outputs spread(std) {ref.std(0)}  mean {ref.mean(0)}")
    def report(tag, out):
        e = out - ref; de = dm(out) - dm(ref)
        sp = ref.std(0)
        corr = [np.corrcoef(out[:, j], ref[:, j])[0, 1] for j in range(ref.shape[1])]
        line = f"{tag:46s} common-mode |mean err|/spread {np.abs(e.mean(0) / sp).max():6.2f}  de-meaned rms/spread {np.sqrt((de**2).mean(0)).__truediv__(sp).max()*100:6.1f}%  max/spread {(np.abs(de).max(0)/sp).max()*100:6.1f}%  corr min {min(corr):.3f}"
        if kind == "refiner":
            dt = np.linalg.norm(e[:, :3], axis=1) * mesh.diameter / 2 * 1e3
            dr = np.degrees(np.linalg.norm(np.tanh(out[:, 3:]) * 0.349065850398865 - np.tanh(ref[:, 3:]) * 0.349065850398865, axis=1))
            line += f"  pose err mm p95 {np.percentile(dt,95):.2f} max {dt.max():.2f}  deg p95 {np.percentile(dr,95):.3f} max {dr.max():.3f}"
        else:
            line += f"  argmax {out[:,0].argmax()} vs {ref[:,0].argmax()} rank-of-winner {list(np.argsort(-ref[:,0])).index(out[:,0].argmax())}"
        print(line, flush=True)
    report(f"{HALF} everywhere (trunk storage + weights; heads fp32)", f16)
    def run(mask, stream_f16, bc, tag, tokfix=False):
        with torch.no_grad():
            t = Trunk(folded, mask, stream_f16, amax)
            if bc:
                # sequential bias correction on the calibration batch: layer by layer, match the per-channel mean of the
                # pre-activation output to the exact network's
                for it in range(bc):
                    if JACOBI:
                        t.record = {}
                        t.forward(Ac, Bc)
                    for i, name in enumerate(LAYERS):
                        if not mask[i]: continue
                        if not JACOBI:
                            t.record = {}
                            t.forward(Ac, Bc)
                        key = ("post:" + name) if POSTBC else name
                        d = (rec_exact[key] - t.record[key]) * (POSTGAIN if POSTBC else 1.0)
                        t.bias_fix[name] = t.bias_fix.get(name, 0) + d
                    t.record = None
            if tokfix:
                t.tokfix = fc.mean(dim=(0, 2, 3)) - t.forward(Ac, Bc).mean(dim=(0, 2, 3))
                tag += ", token corr"
            report(tag, heads(net, kind, t.forward(At, Bt)).numpy())
    ALL = [1] * 13
    which = os.environ.get("RUNS", "base,f16s,bc,f16s_bc,groups").split(",")
    if "base" in which: run(ALL, False, 0, "fp8 all 13, fp8 stream (round 3)")
    if "f16s" in which: run(ALL, True, 0, "fp8 all 13, f16 stream")
    if "bc" in which: run(ALL, False, 1, "fp8 all 13, fp8 stream, bias corr")
    if "f16s_bc" in which: run(ALL, True, 1, "fp8 all 13, f16 stream, bias corr")
    if "tok" in which:
        run(ALL, True, 0, f"{MODE} all 13, f16 stream", True)
        run(ALL, True, SWEEPS, f"{MODE} all 13, f16 stream, bias corr x{SWEEPS} post={POSTBC}", True)
    if "groups" in which:
        G = {"128": range(0, 4), "256": range(4, 8), "b2": range(8, 9), "512": range(9, 13)}
        for k, r in G.items():
            m = [1 if i in r else 0 for i in range(13)]
            run(m, True, 0, f"fp8 only {k}, f16 stream")
            run(m, True, 1, f"fp8 only {k}, f16 stream, bias corr")
        for ks in (("128", "256"), ("128", "256", "b2"), ("256", "b2", "512")):
            m = [1 if any(i in G[k] for k in ks) else 0 for i in range(13)]
            run(m, True, 1, f"fp8 {'+'.join(ks)}, f16 stream, bias corr")


if __name__ == "__main__":
    main()
