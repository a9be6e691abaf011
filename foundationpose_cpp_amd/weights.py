"""Weight ingestion for the refine-net / score-net (replaces tools/cvt_onnx2trt.bash of the reference, which turns
the Google-Drive ONNX files into TensorRT engines; neither file is available offline).

* `make_synthetic_state(kind, seed)`  seeded stand-in weights of the published architecture,
  He-normal convs / Xavier linears / non-trivial BatchNorm statistics.  Every report using them says "synthetic weights".
* `fold_batchnorm(state)`              conv+BN -> conv weight/bias (what TensorRT does when it builds the engine).
* `write_fpw / read_fpw`               the packed "FPW1" container the C library loads (fp32 tensors, PyTorch layouts).
* `python -m foundationpose_cpp_amd.weights --onnx refiner refiner_hwc.onnx refiner.fpw`
                                       real weights: onnx_reader.py pulls the initialisers out of the ONNX files.

numpy only.
"""
from __future__ import annotations

import struct

import numpy as np

BN_EPS = 1e-5
EMBED = 512
HEADS = 4

_RES = "res"
_CBR = "cbr"
ENCODE_A = [(_CBR, 6, 64, 7, 2), (_CBR, 64, 128, 3, 2), (_RES, 128), (_RES, 128)]
ENCODE_AB = [(_RES, 256), (_RES, 256), (_CBR, 256, 512, 3, 2), (_RES, 512), (_RES, 512)]


def _conv(rng, cout, cin, k):
    fan_in = cin * k * k
    return (rng.normal(0, np.sqrt(2.0 / fan_in), size=(cout, cin, k, k)).astype(np.float32),
            rng.normal(0, 0.01, size=cout).astype(np.float32))


def _bn(rng, c):
    return dict(weight=rng.uniform(0.9, 1.1, c).astype(np.float32), bias=rng.normal(0, 0.02, c).astype(np.float32),
                running_mean=rng.normal(0, 0.02, c).astype(np.float32),
                running_var=rng.uniform(0.9, 1.1, c).astype(np.float32))


def _linear(rng, out, inp):
    b = np.sqrt(6.0 / (inp + out))
    return rng.uniform(-b, b, size=(out, inp)).astype(np.float32), rng.normal(0, 0.01, size=out).astype(np.float32)


def _encoder(rng, st, prefix, spec):
    for i, layer in enumerate(spec):
        if layer[0] == _CBR:
            _, cin, cout, k, _s = layer
            st[f"{prefix}.{i}.conv.weight"], st[f"{prefix}.{i}.conv.bias"] = _conv(rng, cout, cin, k)
            for kk, v in _bn(rng, cout).items():
                st[f"{prefix}.{i}.bn.{kk}"] = v
        else:
            c = layer[1]
            for j in (1, 2):
                # second conv of a residual branch scaled down so the residual stack keeps O(1) activations
                w, b = _conv(rng, c, c, 3)
                st[f"{prefix}.{i}.conv{j}.weight"], st[f"{prefix}.{i}.conv{j}.bias"] = (w * (0.5 if j == 2 else 1.0)), b
                for kk, v in _bn(rng, c).items():
                    st[f"{prefix}.{i}.bn{j}.{kk}"] = v


def _mha(rng, st, prefix):
    st[f"{prefix}.in_proj_weight"], st[f"{prefix}.in_proj_bias"] = _linear(rng, 3 * EMBED, EMBED)
    st[f"{prefix}.out_proj.weight"], st[f"{prefix}.out_proj.bias"] = _linear(rng, EMBED, EMBED)


def make_synthetic_state(kind: str, seed: int = 7, calibration: dict | None = None) -> dict:
    """Seeded stand-in weights.  `calibration` (see `apply_calibration`) turns the plain draws into the DISCRIMINATING set the
    parity tests use: same draws, re-centred / re-scaled stage by stage so that differences between hypotheses reach the
    outputs (tests/golden/disc_calib_seed9.npz, produced by oracle/disc_weights.py)."""
    assert kind in ("refiner", "scorer")
    rng = np.random.default_rng(seed + (0 if kind == "refiner" else 1000))
    st: dict = {}
    _encoder(rng, st, "encodeA", ENCODE_A)
    _encoder(rng, st, "encodeAB", ENCODE_AB)
    if kind == "refiner":
        for head, odim in (("trans_head", 3), ("rot_head", 3)):
            _mha(rng, st, f"{head}.0.self_attn")
            st[f"{head}.0.linear1.weight"], st[f"{head}.0.linear1.bias"] = _linear(rng, EMBED, EMBED)
            st[f"{head}.0.linear2.weight"], st[f"{head}.0.linear2.bias"] = _linear(rng, EMBED, EMBED)
            for n in ("norm1", "norm2"):
                st[f"{head}.0.{n}.weight"] = rng.uniform(0.9, 1.1, EMBED).astype(np.float32)
                st[f"{head}.0.{n}.bias"] = rng.normal(0, 0.02, EMBED).astype(np.float32)
            # small output layer: refinement deltas stay a few mm / degrees, so the object stays inside the score crop
            w, b = _linear(rng, odim, EMBED)
            st[f"{head}.1.weight"], st[f"{head}.1.bias"] = w * 0.05, b
    else:
        _mha(rng, st, "att")
        _mha(rng, st, "att_cross")
        st["linear.weight"], st["linear.bias"] = _linear(rng, 1, EMBED)
    if calibration is not None:
        apply_calibration(st, kind, calibration)
    return st


def apply_calibration(st: dict, kind: str, calibration: dict) -> dict:
    """Apply a calibration record to a state dict in place.  Keys are prefixed with the network kind:
      "<kind>/<tensor name>"          replaces that tensor (BatchNorm running statistics, attention biases, output layers),
      "<kind>/gain:<mha prefix>"      multiplies the W_q / W_k rows (first 2*EMBED) of `<mha prefix>.in_proj_weight`."""
    for key, val in calibration.items():
        k, _, name = key.partition("/")
        if k != kind:
            continue
        val = np.asarray(val, np.float32)
        if name.startswith("gain:"):
            w = st[name[5:] + ".in_proj_weight"]
            w[:2 * EMBED] = w[:2 * EMBED] * np.float32(val)
        else:
            assert name in st and st[name].shape == val.shape, (name, val.shape)
            st[name] = val.copy()
    return st


def load_calibration(path: str) -> dict:
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def fold_batchnorm(state: dict) -> dict:
    """conv -> BN(eval) == conv with w*g/sqrt(var+eps), (b-mean)*g/sqrt(var+eps)+beta."""
    out = {}
    done = set()
    for k, w in state.items():
        parts = k.split(".")
        if parts[-1] != "weight" or not parts[-2].startswith("conv"):
            continue
        conv = ".".join(parts[:-1])
        bn = ".".join(parts[:-2] + [parts[-2].replace("conv", "bn")])   # conv->bn, conv1->bn1, conv2->bn2
        b = state[conv + ".bias"]
        g, beta = state[bn + ".weight"], state[bn + ".bias"]
        mu, var = state[bn + ".running_mean"], state[bn + ".running_var"]
        sc = (g.astype(np.float64) / np.sqrt(var.astype(np.float64) + BN_EPS))
        name = ".".join(parts[:-2]) if parts[-2] == "conv" else conv
        out[name + ".weight"] = (w.astype(np.float64) * sc[:, None, None, None]).astype(np.float32)
        out[name + ".bias"] = ((b.astype(np.float64) - mu) * sc + beta).astype(np.float32)
        done.update({conv + ".weight", conv + ".bias"} | {bn + "." + s for s in ("weight", "bias", "running_mean", "running_var")})
    for k, v in state.items():
        if k not in done and "num_batches_tracked" not in k:
            out[k] = np.asarray(v, np.float32)
    return out


def write_fpw(path: str, tensors: dict) -> None:
    with open(path, "wb") as f:
        f.write(b"FPW1")
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", a.ndim))
            f.write(struct.pack("<%dI" % a.ndim, *a.shape))
            f.write(struct.pack("<Q", a.nbytes))
            f.write(a.tobytes())


def read_fpw(path: str) -> dict:
    out = {}
    with open(path, "rb") as f:
        assert f.read(4) == b"FPW1"
        (n,) = struct.unpack("<I", f.read(4))
        for _ in range(n):
            (ln,) = struct.unpack("<I", f.read(4))
            name = f.read(ln).decode()
            (nd,) = struct.unpack("<I", f.read(4))
            shape = struct.unpack("<%dI" % nd, f.read(4 * nd))
            (nbytes,) = struct.unpack("<Q", f.read(8))
            out[name] = np.frombuffer(f.read(nbytes), dtype=np.float32).reshape(shape).copy()
    return out


def pack_synthetic(kind: str, path: str, seed: int = 7, calibration: dict | None = None) -> dict:
    st = make_synthetic_state(kind, seed, calibration)
    write_fpw(path, fold_batchnorm(st))
    return st


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="pack refine-net / score-net weights into the FPW1 container")
    ap.add_argument("--synthetic", nargs=2, metavar=("KIND", "OUT"), help="KIND = refiner|scorer")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--onnx", nargs=3, metavar=("KIND", "ONNX", "OUT"),
                    help="read the initialisers of refiner_hwc.onnx / scorer_hwc.onnx (reference tools/cvt_onnx2trt.bash step)")
    ap.add_argument("--list", metavar="ONNX", help="print what the ONNX reader sees in a file")
    a = ap.parse_args()
    if a.synthetic:
        pack_synthetic(a.synthetic[0], a.synthetic[1], a.seed)
        print("wrote", a.synthetic[1])
    if a.onnx:
        from . import onnx_reader
        st = onnx_reader.convert(a.onnx[1], a.onnx[0], a.onnx[2])
        print("wrote", a.onnx[2], f"({len(st)} tensors)")
    if a.list:
        from . import onnx_reader
        print(onnx_reader.describe(a.list))
