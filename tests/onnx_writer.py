"""Minimal ONNX (protobuf) writer -- TEST INFRASTRUCTURE for foundationpose_cpp_amd/onnx_reader.py.

The real refiner_hwc.onnx / scorer_hwc.onnx are not available offline and neither `onnx` nor torch's exporter can run
in this image, so the reader is exercised on graphs written here in the conventions of the TorchScript exporter [EXT]:
  flavour "folded":  BatchNorm folded into Conv, anonymous `onnx::Conv_N` / transposed `onnx::MatMul_N` constants,
                     raw_data tensors, MatMul + Add linears, LayerNormalization (opset 17), packed in_proj;
  flavour "named":   Conv + BatchNormalization nodes with module-named initialisers, Gemm(transB=1) linears,
                     float_data tensors, q/k/v projections as three MatMuls, the two heads' nodes interleaved.
The graphs carry every weight-bearing node with correct connectivity; the attention / reshape plumbing between them is
schematic (the reader never executes the graph).
"""
import struct

import numpy as np

from foundationpose_cpp_amd import weights as W


def _vi(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(fn, payload):
    return _vi((fn << 3) | 2) + _vi(len(payload)) + bytes(payload)


def _iv(fn, v):
    return _vi(fn << 3) + _vi(v)


def tensor(name, arr, raw=True, half=False):
    if half:     # fp16 initialiser (data_type 10), raw bytes
        arr = np.ascontiguousarray(arr, np.float16)
        return b"".join(_iv(1, d) for d in arr.shape) + _iv(2, 10) + _ld(9, arr.tobytes()) + _ld(8, name.encode())
    arr = np.ascontiguousarray(arr, np.float32)
    b = b"".join(_iv(1, d) for d in arr.shape) + _iv(2, 1)
    b += _ld(9, arr.tobytes()) if raw else _ld(4, arr.tobytes())
    return b + _ld(8, name.encode())


def attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_iv(8, v) for v in vals) + _iv(20, 7)


def attr_i(name, v):
    return _ld(1, name.encode()) + _iv(3, v) + _iv(20, 2)


def attr_f(name, v):
    return _ld(1, name.encode()) + _vi((2 << 3) | 5) + struct.pack("<f", v) + _iv(20, 1)


def node(op, inputs, outputs, attrs=()):
    b = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    b += _ld(3, (op + "_" + outputs[0]).encode()) + _ld(4, op.encode())
    return b + b"".join(_ld(5, a) for a in attrs)


class Builder:
    def __init__(self, flavour):
        self.f, self.nodes, self.inits, self.k = flavour, [], [], 100

    def fresh(self, stem="t"):
        self.k += 1
        return f"/{stem}_{self.k}"

    def const(self, name, arr):
        if self.f == "variant":
            # an exporter without constant folding, fp16 weights: anonymous fp16 initialiser -> Cast(to float) -> consumer
            self.k += 1
            anon = f"val_{self.k}"
            self.inits.append(tensor(anon, arr, half=True))
            return self.add("Cast", [anon], [attr_i("to", 1)])
        self.inits.append(tensor(name, arr, raw=(self.f == "folded")))
        return name

    def add(self, op, inputs, attrs=(), out=None):
        out = out or self.fresh(op)
        self.nodes.append(node(op, inputs, [out], attrs))
        return out

    def conv(self, x, st, base, conv, bn, relu=True):
        w, b = st[f"{base}.{conv}.weight"], st[f"{base}.{conv}.bias"]
        if self.f in ("folded", "variant"):
            fs = W.fold_batchnorm({f"{base}.{conv}.weight": w, f"{base}.{conv}.bias": b,
                                   **{f"{base}.{bn}.{k}": st[f"{base}.{bn}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}})
            (wn, wf), (bn_, bf) = sorted(fs.items(), key=lambda kv: kv[0].endswith("bias"))
            self.k += 2
            y = self.add("Conv", [x, self.const(f"onnx::Conv_{self.k}", wf), self.const(f"onnx::Conv_{self.k + 1}", bf)])
        else:
            y = self.add("Conv", [x, self.const(f"{base}.{conv}.weight", w), self.const(f"{base}.{conv}.bias", b)])
            y = self.add("BatchNormalization", [y] + [self.const(f"{base}.{bn}.{k}", st[f"{base}.{bn}.{k}"])
                                                      for k in ("weight", "bias", "running_mean", "running_var")],
                         [attr_f("epsilon", 1e-5)])
        return self.add("Relu", [y]) if relu else y

    def encoder(self, x, st, prefix, spec):
        for i, layer in enumerate(spec):
            base = f"{prefix}.{i}"
            if layer[0] == "cbr":
                x = self.conv(x, st, base, "conv", "bn")
            else:
                y = self.conv(x, st, base, "conv1", "bn1")
                y = self.conv(y, st, base, "conv2", "bn2", relu=False)
                x = self.add("Relu", [self.add("Add", [y, x])])
        return x

    def linear(self, x, w, b, wname, bname):
        if self.f == "variant":      # weight stored [out, in] behind a Transpose node, bias added on the other side
            y = self.add("MatMul", [x, self.add("Transpose", [self.const(wname, w)], [attr_ints("perm", [1, 0])])])
            return self.add("Add", [y, self.const(bname, b)])
        if self.f == "folded":
            self.k += 1
            y = self.add("MatMul", [x, self.const(f"onnx::MatMul_{self.k}", np.ascontiguousarray(w.T))])
            return self.add("Add", [self.const(bname, b), y])
        return self.add("Gemm", [x, self.const(wname, w), self.const(bname, b)], [attr_i("transB", 1)])

    def mha(self, x, st, prefix):
        w, b = st[f"{prefix}.in_proj_weight"], st[f"{prefix}.in_proj_bias"]
        if self.f == "variant":      # onnxruntime's fused attention: weights [in, 3 * hidden]
            o = self.add("Attention", [x, self.const("w", np.ascontiguousarray(w.T)), self.const("b", b)], [attr_i("num_heads", 4)])
            return self.linear(o, st[f"{prefix}.out_proj.weight"], st[f"{prefix}.out_proj.bias"], f"{prefix}.out_proj.weight", f"{prefix}.out_proj.bias")
        if self.f == "folded":
            qkv = self.linear(x, w, b, f"{prefix}.in_proj_weight", f"{prefix}.in_proj_bias")
            q, k, v = (self.add("Slice", [qkv]) for _ in range(3))
        else:                                               # three separate projections
            E = W.EMBED
            q, k, v = (self.linear(x, w[i * E:(i + 1) * E], b[i * E:(i + 1) * E], f"{prefix}.w{i}", f"{prefix}.b{i}") for i in range(3))
        s = self.add("Softmax", [self.add("MatMul", [q, self.add("Transpose", [k])])])
        o = self.add("MatMul", [s, v])
        return self.linear(o, st[f"{prefix}.out_proj.weight"], st[f"{prefix}.out_proj.bias"], f"{prefix}.out_proj.weight", f"{prefix}.out_proj.bias")

    def layernorm(self, x, st, base):
        if self.f == "variant":      # decomposed, anonymous constants
            mu = self.add("ReduceMean", [x])
            d = self.add("Sub", [x, mu])
            var = self.add("ReduceMean", [self.add("Pow", [d, self.const("two", np.array(2.0, np.float32))])])
            nrm = self.add("Div", [d, self.add("Sqrt", [self.add("Add", [var, self.const("eps", np.array(1e-5, np.float32))])])])
            return self.add("Add", [self.add("Mul", [nrm, self.const("g", st[base + ".weight"])]), self.const("b", st[base + ".bias"])])
        return self.add("LayerNormalization", [x, self.const(base + ".weight", st[base + ".weight"]), self.const(base + ".bias", st[base + ".bias"])],
                        [attr_f("epsilon", 1e-5), attr_i("axis", -1)])

    def model(self, outputs):
        g = b"".join(_ld(1, n) for n in self.nodes) + _ld(2, b"main_graph") + b"".join(_ld(5, t) for t in self.inits)
        g += b"".join(_ld(11, _ld(1, n.encode())) for n in ("render_input", "transf_input"))
        g += b"".join(_ld(12, _ld(1, n.encode())) for n in outputs)
        return _iv(1, 8) + _ld(2, b"pytorch") + _ld(7, g) + _ld(8, _iv(2, 17))


def _pe():
    E = W.EMBED
    pos = np.arange(400, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, E, 2, dtype=np.float32) * np.float32(-(np.log(10000.0) / E)))[None]
    pe = np.zeros((1, 400, E), np.float32)
    pe[0, :, 0::2], pe[0, :, 1::2] = np.sin(pos * div), np.cos(pos * div)
    return pe


def write_model(path, kind, state, flavour="folded"):
    """state: the raw (conv + bn) state dict of weights.make_synthetic_state"""
    b = Builder(flavour)
    x = b.add("Concat", [b.add("Transpose", ["render_input"]), b.add("Transpose", ["transf_input"])], [attr_i("axis", 0)])
    x = b.encoder(x, state, "encodeA", W.ENCODE_A)
    x = b.add("Concat", [b.add("Slice", [x]), b.add("Slice", [x])], [attr_i("axis", 1)])
    x = b.encoder(x, state, "encodeAB", W.ENCODE_AB)
    pe_name = "pos_embed.pe"
    b.inits.append(tensor(pe_name, _pe()))
    x = b.add("Add", [b.add("Transpose", [b.add("Reshape", [x])]), pe_name])
    if kind == "refiner":
        def head_steps(head):
            p = f"{head}.0"
            y = yield
            a = b.mha(y, state, f"{p}.self_attn"); yield
            y = b.layernorm(b.add("Add", [y, a]), state, f"{p}.norm1"); yield
            h = b.add("Relu", [b.linear(y, state[f"{p}.linear1.weight"], state[f"{p}.linear1.bias"], f"{p}.linear1.weight", f"{p}.linear1.bias")]); yield
            h = b.linear(h, state[f"{p}.linear2.weight"], state[f"{p}.linear2.bias"], f"{p}.linear2.weight", f"{p}.linear2.bias"); yield
            y = b.layernorm(b.add("Add", [y, h]), state, f"{p}.norm2"); yield
            o = b.linear(y, state[f"{head}.1.weight"], state[f"{head}.1.bias"], f"{head}.1.weight", f"{head}.1.bias")
            b.add("ReduceMean", [o], out="trans" if head == "trans_head" else "rot"); yield
        gens = [head_steps("trans_head"), head_steps("rot_head")]
        for gen in gens:
            next(gen); gen.send(x)
        if flavour == "named":                              # interleave the two heads' nodes
            for _ in range(5):
                for gen in reversed(gens):
                    next(gen)
        else:
            for gen in gens:
                for _ in range(5):
                    next(gen)
        out = ["trans", "rot"]
    else:
        a = b.mha(x, state, "att")
        f = b.add("Unsqueeze", [b.add("ReduceMean", [a])])
        c = b.mha(f, state, "att_cross")
        b.add("Reshape", [b.linear(c, state["linear.weight"], state["linear.bias"], "linear.weight", "linear.bias")], out="scores")
        out = ["scores"]
    open(path, "wb").write(b.model(out))
