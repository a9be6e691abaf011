"""bf16 and FP8 (OCP e4m3) paths: BASELINE.json configs[1] (Track, bf16 refine-net) and configs[4] (1280x720, textured +
untextured, N = 252, FP8 convolutions).

Kernel level (test build, fpt_conv_dt): every schedule an FP8 / bf16 layer can reach against a torch fp32 reference evaluated on the
SAME quantised operands -- products of e4m3 / bf16 values are exact in fp32, so what is left is summation order and the
rounding of the output.  End to end: poses against the f16 path and the oracle pipeline, tolerances stated per test.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from foundationpose_cpp_amd import FoundationPose, _lib, synthetic as syn, weights as W
from foundationpose_cpp_amd.api import FP_PREC_BF16, FP_PREC_F16, FP_PREC_FP8, FP_PREC_INT8
from oracle import fp_oracle as fo
from oracle import nets_torch as NT

pytestmark = pytest.mark.gpu

DT_F16, DT_BF16, DT_FP8, DT_I8 = 0, 1, 2, 3


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---- e4m3 (OCP fn: bias 7, max 448, no inf), round to nearest even, saturating: the numpy twin of f32_to_e4m3_bits ----
def q_e4m3(x):
    x = np.asarray(x, np.float64)
    a = np.minimum(np.abs(x), 448.0)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.maximum(e, -6.0)                      # subnormals share the exponent of the smallest normal
    quantum = 2.0 ** (e - 3)
    q = np.rint(a / quantum) * quantum           # np.rint rounds half to even
    return (np.sign(x) * np.minimum(q, 448.0)).astype(np.float32)


def q_bf16(x):
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def q_f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def _conv_dt(x, w_oihw, bias, stride, pad, relu, res, dt, out_dt, in_scale=1.0, res_scale=1.0, out_scale=1.0):
    L = _lib.test_lib()
    L.fpt_conv_dt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 13 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                                  C.c_float, C.c_float, C.c_float]
    NB, H, Wd, Cin = x.shape
    Cout, _, KH, KW = w_oihw.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (Wd + 2 * pad - KW) // stride + 1
    wk = np.ascontiguousarray(w_oihw.transpose(0, 2, 3, 1), np.float32)
    out = np.zeros((NB, OH, OW, Cout), np.float32)
    x = np.ascontiguousarray(x, np.float32)
    b = np.ascontiguousarray(bias, np.float32)
    r = np.ascontiguousarray(res, np.float32) if res is not None else None
    rc = L.fpt_conv_dt(_p(x), _p(wk), _p(b), _p(r), NB, H, Wd, Cin, Cout, KH, KW, stride, pad, OH, OW, int(relu), 0,
                       _p(out), 1, None, dt, out_dt, in_scale, res_scale, out_scale)
    assert rc == 0, L.fp_last_error()
    return out


def _ref(xq, wq, bias, stride, pad, relu, resq):
    y = torch.nn.functional.conv2d(torch.from_numpy(xq).permute(0, 3, 1, 2).double(), torch.from_numpy(wq).double(),
                                   torch.from_numpy(np.asarray(bias, np.float64)), stride, pad).permute(0, 2, 3, 1)
    if resq is not None:
        y = y + torch.from_numpy(resq).double()
    if relu:
        y = torch.relu(y)
    return y.float().numpy()


# ---- the 8-bit layers (FP8 e4m3 / INT8), exactly as the 8-bit trunk runs them (fpt_conv_q8 = net_apply_q8 + run_conv) ----
def q_act(x, s, dt):
    """what the device holds for activation x with per-channel scales s: e4m3(x / s) * s, or clamp(rint(x / s), 0, 255) * s"""
    if dt == DT_FP8:
        return q_e4m3(x / s) * s
    # (the quotient in float32, like the producer's epilogue: value * (1 / s) differs from it in the last bit only)
    return (np.clip(np.rint(np.asarray(x, np.float32) / np.asarray(s, np.float32)), 0, 255).astype(np.float64) * s).astype(np.float32)


def _conv_q8(x, s_in, w_oihw, bias, stride, relu, res, mode, s_out, dt, split=0, s_res=None):
    L = _lib.test_lib()
    L.fpt_conv_q8.argtypes = [C.c_void_p] * 5 + [C.c_int] * 14 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    NB, H, Wd, Cin = x.shape
    Cout, _, KH, KW = w_oihw.shape
    OH = (H + 2 - KH) // stride + 1
    wk = np.ascontiguousarray(w_oihw.transpose(0, 2, 3, 1), np.float32)
    oshape = (NB - split, OH, OH, 2 * Cout) if split else (NB, OH, OH, Cout)
    o16, oq = np.zeros(oshape, np.float32), np.zeros(oshape, np.float32)
    wq = np.zeros_like(wk)
    x = np.ascontiguousarray(x, np.float32)
    r = np.ascontiguousarray(res, np.float32) if res is not None else None
    rc = L.fpt_conv_q8(_p(x), _p(np.ascontiguousarray(s_in, np.float32)), _p(wk), _p(np.ascontiguousarray(bias, np.float32)), _p(r),
                       NB, H, Wd, Cin, Cout, KH, KW, stride, 1, OH, OH, int(relu), split, mode,
                       _p(np.ascontiguousarray(s_out, np.float32)), _p(o16), _p(oq), 1, None, dt, _p(wq),
                       _p(np.ascontiguousarray(s_res, np.float32)) if s_res is not None else None)
    assert rc == 0, L.fp_last_error()
    return o16, oq, np.ascontiguousarray(wq.transpose(0, 3, 1, 2))


Q8_SHAPES = [
    # NB, H, Cin, Cout, stride, use_res, what it reaches
    (3, 40, 128, 128, 1, True),     # small batch: conv_smallx_kernel
    (70, 40, 128, 128, 1, True),    # conv_halo8_kernel, one 128-channel chunk, residual
    (40, 40, 256, 256, 1, True),    # conv_halo8_kernel, two chunks (halo refill), two channel tiles
    (170, 20, 512, 512, 1, True),   # conv_big_pp_kernel rounds + conv_deep_kernel<64> left-over
    (300, 20, 512, 512, 1, False),  # conv_big_pp rounds + 256x128 ping-pong cascade + deep kernel
    (70, 40, 256, 512, 2, False),   # encodeAB.2: stride 2 on the 256x256 tile
    (1, 20, 512, 512, 1, True),     # Track-sized
    (12, 40, 256, 256, 1, True),    # a few objects: conv_igemm_kernel<128> / mid-sized paths
]


@pytest.mark.parametrize("dt", [DT_FP8, DT_I8], ids=["fp8", "int8"])
@pytest.mark.parametrize("shape", Q8_SHAPES)
def test_q8_conv_schedules(shape, dt):
    """Every schedule an 8-bit layer can reach, in its three output forms, against a float64 reference on the SAME quantised operands
    (products of e4m3 values / of 8-bit integers are exact in fp32 / int32): what is left is the fp32 summation order (FP8; INT8
    accumulates exactly) and the rounding of the outputs."""
    NB, H, Cin, Cout, stride, use_res = shape
    rng = np.random.default_rng(3)
    x = np.maximum(rng.normal(size=(NB, H, H, Cin)), 0).astype(np.float32)          # post-ReLU like the real activations
    x *= rng.uniform(0.3, 3.0, Cin).astype(np.float32)                               # channels of different ranges
    w = (rng.normal(size=(Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = (0.1 * rng.normal(size=Cout)).astype(np.float32)
    OH = (H + 2 - 3) // stride + 1
    res = np.maximum(rng.normal(size=(NB, OH, OH, Cout)), 0).astype(np.float32) if use_res else None
    amax_in = x.reshape(-1, Cin).max(0)
    s_in = (amax_in / 224 if dt == DT_FP8 else amax_in * 1.25 / 255).astype(np.float32)
    s_out = (rng.uniform(4.0, 8.0, Cout) / (224 if dt == DT_FP8 else 255)).astype(np.float32)
    xq = q_act(x, s_in, dt)
    rq = q_f16(res) if use_res else None
    # mode 1 first: it also returns the weights the device multiplies with
    got16, _, wq = _conv_q8(x, s_in, w, b, stride, True, res, 1, s_out, dt)
    wf, ws = w * s_in[None, :, None, None], np.abs(w * s_in[None, :, None, None]).reshape(Cout, -1).max(1)
    # sanity: it IS w, quantised per row after the activation scales were folded in (half a step of the row's range)
    assert np.all(np.abs(wq * s_in[None, :, None, None] - wf).reshape(Cout, -1).max(1) <= ws * (2.0 ** -4 if dt == DT_FP8 else 0.5 / 127) * 1.01)
    ref = _ref(xq, wq, b, stride, 1, True, rq)
    np.testing.assert_allclose(got16, ref, rtol=3e-3, atol=3e-3)                     # f16 output (the trunk's last layer)
    # mode 2: f16 stream tensor + its 8-bit copy
    got16b, gotq, _ = _conv_q8(x, s_in, w, b, stride, True, res, 2, s_out, dt)
    assert np.array_equal(got16b, got16)
    want = q_act(ref, s_out, dt)
    def close(got):   # summation order / the f32 rounding of value * (1 / s) can move a value across a rounding boundary: one step
        assert np.mean(got == want) > 0.99, np.mean(got == want)
        if dt == DT_FP8:
            # one e4m3 step (2^-3 relative; 2^-9 * scale in the subnormal range) -- for all but a few outputs per 10^7: where the sum cancels
            # to ~0 the matrix pipe's 128-wide dot product (not an exact fp32 sum) shows, 2-3 subnormal steps
            viol = np.abs(got - want) > 0.13 * np.abs(want) + float(s_out.max()) * 2.0 ** -9 * 1.01
            assert viol.mean() < 2e-6 and np.abs(got - want)[viol].max(initial=0.0) < float(s_out.max()) * 2.0 ** -9 * 4, (viol.sum(), viol.size)
        else: assert np.all(np.abs(got - want) <= s_out * 1.01)
    close(gotq)
    _, gotq3, _ = _conv_q8(x, s_in, w, b, stride, True, res, 3, s_out, dt)       # mode 3: the scaled 8-bit copy alone
    assert np.array_equal(gotq3, gotq)
    if not use_res:  # mode 0 (a block's first conv): 8-bit only, the consumer's scales folded into the epilogue tables
        _, gotq0, _ = _conv_q8(x, s_in, w, b, stride, True, None, 0, s_out, dt)
        close(gotq0)
    elif dt == DT_I8:  # modes 4 / 5 (the INT8 trunk's 8-bit residual stream): the skip operand is an unsigned 8-bit tensor with its own scales
        s_res = (res.reshape(-1, Cout).max(0) * 1.25 / 255).astype(np.float32)
        ref_rq = _ref(xq, wq, b, stride, 1, True, q_act(res, s_res, dt))
        got16r, _, _ = _conv_q8(x, s_in, w, b, stride, True, res, 5, s_out, dt, s_res=s_res)
        np.testing.assert_allclose(got16r, ref_rq, rtol=3e-3, atol=3e-3)
        _, gotq4, _ = _conv_q8(x, s_in, w, b, stride, True, res, 4, s_out, dt, s_res=s_res)
        want = q_act(ref_rq, s_out, dt)
        close(gotq4)


@pytest.mark.parametrize("dt", [DT_FP8, DT_I8], ids=["fp8", "int8"])
def test_q8_concat_layer(dt):
    """The last encodeA conv writes the a|b channel concat -- both the f16 tensor and its 8-bit copy, one scale table for both halves."""
    rng = np.random.default_rng(8)
    NB, split = 60, 30
    x = np.maximum(rng.normal(size=(NB, 40, 40, 128)), 0).astype(np.float32)
    w = (rng.normal(size=(128, 128, 3, 3)) / np.sqrt(128 * 9)).astype(np.float32)
    b = (0.1 * rng.normal(size=128)).astype(np.float32)
    res = np.maximum(rng.normal(size=(NB, 40, 40, 128)), 0).astype(np.float32)
    s_in = (x.reshape(-1, 128).max(0) * (1 / 224 if dt == DT_FP8 else 1.25 / 255)).astype(np.float32)
    s_out = (rng.uniform(4.0, 8.0, 128) / (224 if dt == DT_FP8 else 255)).astype(np.float32)
    _, _, wq = _conv_q8(x[:2], s_in, w, b, 1, True, res[:2], 1, s_out, dt)
    ref = _ref(q_act(x, s_in, dt), wq, b, 1, 1, True, q_f16(res))
    ref_cat = np.concatenate([ref[:split], ref[split:]], -1)
    got16, gotq, _ = _conv_q8(x, s_in, w, b, 1, True, res, 2, s_out, dt, split=split)
    np.testing.assert_allclose(got16, ref_cat, rtol=2e-3, atol=2e-3)
    want = q_act(ref_cat, np.concatenate([s_out, s_out]), dt)
    assert np.mean(gotq == want) > 0.99
    if dt == DT_I8:   # the INT8 trunk's form: 8-bit residual in, scaled 8-bit concat out
        s_res = (res.reshape(-1, 128).max(0) * 1.25 / 255).astype(np.float32)
        ref = _ref(q_act(x, s_in, dt), wq, b, 1, 1, True, q_act(res, s_res, dt))
        want = q_act(np.concatenate([ref[:split], ref[split:]], -1), np.concatenate([s_out, s_out]), dt)
        _, gotq4, _ = _conv_q8(x, s_in, w, b, 1, True, res, 4, s_out, dt, split=split, s_res=s_res)
        assert np.mean(gotq4 == want) > 0.99 and np.all(np.abs(gotq4 - want) <= np.concatenate([s_out, s_out]) * 1.01)


@pytest.mark.parametrize("dt", [DT_FP8, DT_I8], ids=["fp8", "int8"])
@pytest.mark.parametrize("NB", [3, 40])
def test_f16_to_q8_boundary_layer(NB, dt):
    """encodeA.1 in the 8-bit precisions: f16 operands (3x3 / stride 2, 64 -> 128 on 80x80), f16 stream output + 8-bit copy; NB = 40
    reaches conv_s2_halo_kernel, NB = 3 the small-batch path."""
    rng = np.random.default_rng(4)
    x = np.maximum(rng.normal(size=(NB, 80, 80, 64)), 0).astype(np.float32)
    w = (rng.normal(size=(128, 64, 3, 3)) / np.sqrt(64 * 9)).astype(np.float32)
    b = (0.1 * rng.normal(size=128)).astype(np.float32)
    s_out = (rng.uniform(3.0, 6.0, 128) / (224 if dt == DT_FP8 else 255)).astype(np.float32)
    L = _lib.test_lib()
    L.fpt_conv_f16_dual.argtypes = [C.c_void_p] * 3 + [C.c_int] * 10 + [C.c_void_p] * 3 + [C.c_int]
    o16, oq = np.zeros((NB, 40, 40, 128), np.float32), np.zeros((NB, 40, 40, 128), np.float32)
    wk = np.ascontiguousarray(w.transpose(0, 2, 3, 1))
    assert L.fpt_conv_f16_dual(_p(x), _p(wk), _p(b), NB, 80, 80, 64, 128, 3, 3, 2, 1, 40, _p(s_out), _p(o16), _p(oq), dt) == 0, L.fp_last_error()
    ref = _ref(q_f16(x), q_f16(w), b, 2, 1, True, None)
    np.testing.assert_allclose(o16, ref, rtol=2e-3, atol=2e-3)
    want = q_act(ref, s_out, dt)
    assert np.mean(oq == want) > 0.99
    if dt == DT_I8:   # the INT8 trunk writes the 8-bit copy alone
        oq2 = np.zeros_like(oq)
        assert L.fpt_conv_f16_dual(_p(x), _p(wk), _p(b), NB, 80, 80, 64, 128, 3, 3, 2, 1, 40, _p(s_out), None, _p(oq2), dt) == 0, L.fp_last_error()
        assert np.array_equal(oq2, oq)


BF16_SHAPES = [
    # NB, H, W, Cin, Cout, k, stride, use_res
    (3, 40, 40, 128, 128, 3, 1, True),     # igemm
    (70, 40, 40, 128, 128, 3, 1, True),    # conv_halo_kernel<bf16>
    (40, 80, 80, 64, 128, 3, 2, False),    # conv_s2_halo_kernel<bf16>
    (40, 80, 80, 32, 64, 4, 1, False),     # conv_stem_halo_kernel<bf16> (s2d stem mode)
    (170, 20, 20, 512, 512, 3, 1, True),   # 256x256 ping-pong + deep kernel
    (700, 40, 40, 128, 128, 3, 1, False),  # conv_pp32_kernel<512,128> rounds + left-over
    (1, 66001, 1, 512, 1536, 1, 1, False), # gemm_k32_kernel<bf16>
    (1, 400, 1, 512, 512, 1, 1, True),     # Track-sized Linear layer: split-K
]


@pytest.mark.parametrize("shape", BF16_SHAPES)
def test_bf16_conv_schedules(shape):
    NB, H, Wd, Cin, Cout, k, stride, use_res = shape
    rng = np.random.default_rng(5)
    x = rng.normal(size=(NB, H, Wd, Cin)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    pad = 2 if k == 4 else (k - 1) // 2
    stem = k == 4
    OH = H if stem else (H + 2 * pad - k) // stride + 1
    OW = Wd if stem else (Wd + 2 * pad - k) // stride + 1
    res = rng.normal(size=(NB, OH, OW, Cout)).astype(np.float32) if use_res else None
    if stem:   # asymmetric padding (2 before, 1 after) of the space-to-depth stem
        xt = torch.nn.functional.pad(torch.from_numpy(q_bf16(x)).permute(0, 3, 1, 2), (2, 1, 2, 1))
        ref = torch.relu(torch.nn.functional.conv2d(xt, torch.from_numpy(q_bf16(w)), torch.from_numpy(b))).permute(0, 2, 3, 1).numpy()
        L = _lib.test_lib()
        L.fpt_conv_dt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 13 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                                      C.c_float, C.c_float, C.c_float]
        got = np.zeros((NB, OH, OW, Cout), np.float32)
        wk = np.ascontiguousarray(w.transpose(0, 2, 3, 1), np.float32)
        assert L.fpt_conv_dt(_p(x), _p(wk), _p(b), None, NB, H, Wd, Cin, Cout, k, k, 1, 2, OH, OW, 1, 0, _p(got), 1, None,
                             DT_BF16, DT_BF16, 1.0, 1.0, 1.0) == 0, L.fp_last_error()
    else:
        ref = _ref(q_bf16(x), q_bf16(w), b, stride, pad, True, q_bf16(res) if use_res else None)
        got = _conv_dt(x, w, b, stride, pad, True, res, DT_BF16, DT_BF16)
    # output rounded to bf16: 2^-9 relative + summation order
    np.testing.assert_allclose(got, ref, rtol=8e-3, atol=8e-3)


@pytest.mark.parametrize("B,T", [(3, 400), (1, 252)])
def test_bf16_attention(B, T):
    L = _lib.test_lib()
    rng = np.random.default_rng(6)
    qkv = rng.normal(size=(B, T, 1536)).astype(np.float32)
    out = np.zeros((B, T, 512), np.float32)
    assert L.fpt_attention_dt(_p(qkv), B, T, _p(out), DT_BF16) == 0, L.fp_last_error()
    q, k, v = [torch.from_numpy(q_bf16(qkv[..., i * 512:(i + 1) * 512])).view(B, T, 4, 128).transpose(1, 2) for i in range(3)]
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, 512).numpy()
    np.testing.assert_allclose(out, ref, rtol=2e-2, atol=2e-2)   # P is rounded to bf16 (2^-9) before the PV product


# ---------------------------------------------------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def nets(tmp_path_factory):
    d = tmp_path_factory.mktemp("w")
    rp, sp = str(d / "refiner.fpw"), str(d / "scorer.fpw")
    rs = W.pack_synthetic("refiner", rp)
    ss = W.pack_synthetic("scorer", sp)
    return rp, sp, NT.build("refiner", rs), NT.build("scorer", ss)


def _pose_err(a, b):
    dR = a[:3, :3] @ b[:3, :3].T
    ang = np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
    return ang, np.linalg.norm(a[:3, 3] - b[:3, 3])


def _oracle_track(nets, mesh, scene, hyp):
    om = fo.OracleMesh(mesh)
    p16 = syn.to_colmajor(hyp[None])
    a = fo.render(om, p16, scene.K, scene.depth.shape, 1.2)
    b = fo.crop(scene.rgb, scene.depth, scene.K, p16, 1.2, mesh.diameter)
    with torch.no_grad():
        t, r = nets[2](torch.from_numpy(a), torch.from_numpy(b))
    return syn.from_colmajor(fo.refine_post_process(p16, t.numpy(), r.numpy(), mesh.diameter))[0]


def test_track_bf16_refine_net(nets, syn_mesh, syn_scene):
    """BASELINE configs[1]: Track, N = 1, bf16 refine-net.  bf16 keeps 8 significand bits (f16: 11), so the refiner's
    outputs carry ~8x the f16 rounding noise; the pose must still sit well inside the north-star bar (1 deg / 1 mm):
    asserted 0.3 deg / 0.3 mm against the fp32 oracle pipeline and against the f16 path."""
    m = FoundationPose(syn_mesh, syn.intrinsics(), nets[0], nets[1])
    try:
        hyp = syn.perturb_pose(syn_scene.gt_pose)
        ok, p16 = m.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
        assert ok, m.last_error
        m.set_precision(FP_PREC_BF16)
        assert m.precision == FP_PREC_BF16
        poses = []
        for _ in range(3):      # eager, capture, replay
            ok, pb = m.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
            assert ok, m.last_error
            poses.append(pb)
        assert np.array_equal(poses[0], poses[1]) and np.array_equal(poses[1], poses[2])
        ref = _oracle_track(nets, syn_mesh, syn_scene, hyp)
        for other in (ref, p16):
            ang, dist = _pose_err(poses[0], other)
            assert ang < 0.3 and dist < 3e-4, (ang, dist)
        # and a bf16 Register runs end to end and agrees with the f16 one on the winner's pose
        ok, rb = m.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)
        assert ok, m.last_error
        m.set_precision(FP_PREC_F16)
        ok, rh = m.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)
        assert ok
        print("bf16 vs f16 Register pose:", _pose_err(rb, rh))
    finally:
        m.close()


@pytest.mark.parametrize("prec,name", [(FP_PREC_FP8, "fp8"), (FP_PREC_INT8, "int8")])
def test_track_and_small_batches_q8(nets, syn_mesh, syn_scene, prec, name):
    """The small-batch 8-bit schedules (conv_smallx_kernel and the mid-sized paths) that Track and small Register slices take: Track in
    the 8-bit precision against the fp32 oracle pipeline and the f16 path; the calibration record survives a get / set round trip
    into a fresh model (bit-identical results)."""
    m = FoundationPose(syn_mesh, syn.intrinsics(), nets[0], nets[1])
    m2 = FoundationPose(syn_mesh, syn.intrinsics(), nets[0], nets[1])
    try:
        hyp = syn.perturb_pose(syn_scene.gt_pose)
        ok, p16 = m.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
        assert ok, m.last_error
        m.calibrate(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name, prec)
        assert m.precision == FP_PREC_F16                      # calibration leaves the precision alone
        cal = m.get_calibration()
        assert cal.shape == (32,) and np.all(cal[1:14] > 0) and np.all(cal[17:30] > 0)
        blob = m.get_calibration_blob(prec)
        m.set_precision(prec)
        poses = []
        for _ in range(3):      # eager, capture, replay
            ok, p8 = m.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
            assert ok, m.last_error
            poses.append(p8)
        assert np.array_equal(poses[0], poses[1]) and np.array_equal(poses[1], poses[2])
        ref = _oracle_track(nets, syn_mesh, syn_scene, hyp)
        for other in (ref, p16):
            ang, dist = _pose_err(poses[0], other)
            assert ang < 1.0 and dist < 1e-3, (ang, dist)      # the north-star bar (plain weights: a common-mode check)
        print(f"{name} Track vs oracle / f16:", _pose_err(poses[0], ref), _pose_err(poses[0], p16))
        # a fresh model with the stored record reproduces the result bit for bit
        m2.set_calibration_blob(blob)
        m2.set_precision(prec)
        ok, q8 = m2.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
        assert ok and np.array_equal(q8, poses[0])
        with pytest.raises(Exception):
            m2.set_calibration_blob(blob[:-4])
        # a 42-hypothesis Register (one in-plane step): runs the mid-sized schedules end to end
        m.set_inplane_steps(1)
        ok, r8 = m.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)
        assert ok, m.last_error
        m.set_precision(FP_PREC_F16)
        ok, r16 = m.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)
        assert ok
        print(f"{name} vs f16 Register (42 hypotheses) pose:", _pose_err(r8, r16))
    finally:
        m.close()
        m2.close()


def test_q8_needs_calibration(nets, syn_mesh, syn_scene):
    m = FoundationPose(syn_mesh, syn.intrinsics(), nets[0], nets[1])
    try:
        for prec in (FP_PREC_FP8, FP_PREC_INT8):
            with pytest.raises(Exception) as e:
                m.set_precision(prec)
            assert "calibrat" in str(e.value)
            assert m.precision == FP_PREC_F16
        # the round-2 per-tensor interface still works (every channel gets the tensor's scale, no corrections)
        m.calibrate(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name, FP_PREC_FP8)
        with pytest.raises(Exception) as e:      # a calibration belongs to ITS precision (statistics + the corrections solved against them)
            m.set_precision(FP_PREC_INT8)
        assert "calibrat" in str(e.value)
        with pytest.raises(Exception):
            m.get_calibration_blob(FP_PREC_INT8)
        cal = m.get_calibration()
        m2 = FoundationPose(syn_mesh, syn.intrinsics(), nets[0], nets[1])
        try:
            m2.set_calibration(cal)
            m2.set_precision(FP_PREC_FP8)
            ok, _ = m2.Track(syn_scene.rgb, syn_scene.depth, syn.perturb_pose(syn_scene.gt_pose), syn_mesh.name)
            assert ok, m2.last_error
        finally:
            m2.close()
    finally:
        m.close()


def _rot_deg(a, b):
    dR = np.einsum("nij,nkj->nik", a[:, :3, :3].astype(np.float64), b[:, :3, :3].astype(np.float64))
    return np.degrees(np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1) / 2, -1, 1)))


# ---- BASELINE configs[4] (1280x720, textured + untextured mesh, N = 252) -------------------------------------------------------------
# [r6] configs[4] runs in F16.  Its text says "fp8 MFMA conv path"; neither 8-bit precision holds the north-star bar (>= 95 % of the refined
# poses within 1 mm / 1 deg of the 16-bit path on EVERY frame the calibration never saw, common-mode shift < 0.3 mm) under the
# discriminating weights, for ANY subset of trunk stages: tools/q8_blocks.py (profiles/r06_q8_blocks_*.log) ran INT8 on every single
# stage and on the combinations -- the best single stage (encodeAB.2 alone, 4 % of the FLOPs, no speed-up) reaches 100 % with 0.31-0.40 mm
# of common mode, the 256- or 512-channel stage alone 94-100 % / 0.4-0.9 mm for 8-9 % of speed, all 13 layers 85-90 % / 0.7-1.0 mm for 22 %.
# So (round-5 review, item 1): the row is "f16 only"; FP_PREC_INT8 and FP_PREC_FP8 stay selectable, kernel-tested and Track-tested but
# EXPERIMENTAL, and what this section asserts about them is (a) the f16 path itself against the fp32 ORACLE at the configs[4] size -- the
# parity statement of the row -- and (b) regression guards of the experimental precisions at their measured level, named as such, with
# the bar itself recorded as a STRICT xfail for each.
# Calibration: fp_calibrate_begin / _add_frame x 16 / _finish on syn.calibration_scenes (the synthetic "deployment scene family":
# object 0.55-0.95 m away, up to 6 cm off axis, any orientation, own noise / dropped pixels / background); measurement on
# syn.heldout_scenes (same family, other seeds).  Per held-out scene, N = 252:
#   frac     share of the 252 refined poses within 1 mm / 1 deg of the F16 path's refined pose of the same hypothesis
#   cm       common-mode shift: length of the mean translation difference over the 252 hypotheses
#   corr     correlation of the refiner's pose deltas with the f16 path's
#   regret   teacher-forced: the f16 model scores the 8-bit model's refined poses; (best - winner's) / (best - median)
_CAL = {}


def _calibrated_int8_blob(disc_nets, Wd, H, textured, prec=FP_PREC_INT8, k=16):
    """one calibration per (size, mesh, precision) for the whole module: the record travels as a blob (also what a deployment does)"""
    key = (Wd, H, textured, prec, k)
    if key not in _CAL:
        mesh = syn.make_mesh(textured=textured)
        m = FoundationPose(mesh, syn.intrinsics(Wd, H), disc_nets[0], disc_nets[1])
        try:
            m.calibrate_frames(syn.calibration_scenes(mesh, k, W=Wd, H=H), mesh.name, prec)
            assert m.precision == FP_PREC_F16
            _CAL[key] = m.get_calibration_blob(prec)
        finally:
            m.close()
    return _CAL[key]


def _heldout_stats(m, mesh, scene, prec):
    m.set_precision(FP_PREC_F16)
    ok, p16, idx16, sc16, ref16, _ = m.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
    assert ok, m.last_error
    m.upload_frame(scene.rgb, scene.depth)
    hyp = m.get_hyp_poses(scene.mask)
    m.set_precision(prec)
    ok, p8, idx8, sc8, ref8, _ = m.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
    assert ok, m.last_error
    ok, p8b, idx8b, _, _, _ = m.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
    assert ok and idx8b == idx8 and np.array_equal(p8, p8b)          # deterministic
    dmm = np.linalg.norm(ref8[:, :3, 3] - ref16[:, :3, 3], axis=1) * 1e3
    ddeg = _rot_deg(ref8, ref16)
    d8, d16 = ref8[:, :3, 3] - hyp[:, :3, 3], ref16[:, :3, 3] - hyp[:, :3, 3]
    corr = min(np.corrcoef(d8[:, j], d16[:, j])[0, 1] for j in range(3))
    corr = min(corr, np.corrcoef(_rot_deg(ref8, hyp), _rot_deg(ref16, hyp))[0, 1])
    m.set_precision(FP_PREC_F16)
    m.upload_frame(scene.rgb, scene.depth)
    sc_tf = m.scorer_infer(*m.render_and_transform(mesh.name, ref8, 1.1))
    return dict(frac=float(np.mean((dmm < 1.0) & (ddeg < 1.0))), mm_p95=float(np.percentile(dmm, 95)), mm_max=float(dmm.max()),
                deg_p95=float(np.percentile(ddeg, 95)), cm=float(np.linalg.norm((ref8[:, :3, 3] - ref16[:, :3, 3]).mean(0)) * 1e3),
                corr=float(corr), rank=int((sc_tf > sc_tf[idx8]).sum()),
                regret=float((sc_tf.max() - sc_tf[idx8]) / (sc_tf.max() - np.median(sc_tf))), score_corr=float(np.corrcoef(sc8, sc_tf)[0, 1]),
                idx8=idx8, idx16=idx16)


def _heldout_table(disc_nets, Wd, H, textured, prec, n_scenes):
    mesh = syn.make_mesh(textured=textured)
    blob = _calibrated_int8_blob(disc_nets, Wd, H, textured, prec)
    m = FoundationPose(mesh, syn.intrinsics(Wd, H), disc_nets[0], disc_nets[1])
    try:
        m.set_calibration_blob(blob)
        rows = [_heldout_stats(m, mesh, sc, prec) for sc in syn.heldout_scenes(mesh, n_scenes, W=Wd, H=H)]
    finally:
        m.close()
    for k, r in enumerate(rows):
        print(f"  held-out scene {k}: within 1 mm / 1 deg {r['frac'] * 100:5.1f} %, mm p95 {r['mm_p95']:.2f} max {r['mm_max']:.2f} (common-mode {r['cm']:.2f}), "
              f"deg p95 {r['deg_p95']:.2f}, delta corr {r['corr']:.4f}, winner {r['idx8']} (f16 {r['idx16']}): teacher-forced rank {r['rank']} regret {r['regret']:.4f}, "
              f"score corr {r['score_corr']:.4f}")
    return rows


@pytest.mark.parametrize("textured", [True, False])
def test_configs4_f16_720p_follows_the_oracle(disc_nets, textured):
    """BASELINE configs[4] as it ships: 1280x720, textured + untextured mesh, N = 252, F16.  Under the discriminating weights, on two
    held-out scenes: the f16 path's 252 refined poses against the fp32 ORACLE chain (oracle/fp_oracle.c geometry + torch networks on the
    same hypotheses) hold the north-star bar -- >= 95 % within 1 mm / 1 deg, common-mode shift < 0.3 mm -- with a wide margin (measured:
    100 %, p95 ~0.1 mm), and the pose the Register returns is one of the oracle's refined poses whose oracle score is within the f16
    score noise of the oracle's maximum (teacher-forced like every end-to-end comparison here: rendering is discontinuous in the pose)."""
    mesh = syn.make_mesh(textured=textured)
    om = fo.OracleMesh(mesh)
    m = FoundationPose(mesh, syn.intrinsics(1280, 720), disc_nets[0], disc_nets[1])
    try:
        for k, scene in enumerate(syn.heldout_scenes(mesh, 2, W=1280, H=720)):
            ok, pose, idx, scores, refined, _ = m.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name)
            assert ok, m.last_error
            p16 = fo.get_hyp_poses(scene.depth, scene.mask, scene.K)
            a = fo.render(om, p16, scene.K, scene.depth.shape, 1.2)
            b = fo.crop(scene.rgb, scene.depth, scene.K, p16, 1.2, mesh.diameter)
            with torch.no_grad():
                t, r = disc_nets[2](torch.from_numpy(a), torch.from_numpy(b))
            ref = syn.from_colmajor(fo.refine_post_process(p16, t.numpy(), r.numpy(), mesh.diameter))
            dmm = np.linalg.norm(refined[:, :3, 3] - ref[:, :3, 3], axis=1) * 1e3
            ddeg = _rot_deg(refined, ref)
            cm = float(np.linalg.norm((refined[:, :3, 3] - ref[:, :3, 3]).mean(0)) * 1e3)
            frac = float(np.mean((dmm < 1.0) & (ddeg < 1.0)))
            print(f"f16 1280x720 textured={textured} held-out scene {k}: within 1 mm / 1 deg of the fp32 oracle {frac * 100:.1f} %, mm p95 {np.percentile(dmm, 95):.3f} "
                  f"max {dmm.max():.3f} (common-mode {cm:.3f}), deg p95 {np.percentile(ddeg, 95):.3f}")
            assert frac >= 0.95 and cm < 0.3, (frac, cm)
            assert np.percentile(dmm, 95) < 0.5 and np.percentile(ddeg, 95) < 0.5
            # the winner, teacher-forced: the ORACLE scores the library's own refined poses
            r16 = syn.to_colmajor(refined)
            a = fo.render(om, r16, scene.K, scene.depth.shape, 1.1)
            b = fo.crop(scene.rgb, scene.depth, scene.K, r16, 1.1, mesh.diameter)
            with torch.no_grad():
                s_ref = disc_nets[3](torch.from_numpy(a), torch.from_numpy(b)).numpy()
            assert np.array_equal(pose, refined[idx])
            # the scores: on these scenes the discriminating head sits far from its calibration point (scores 26-27 with a spread of 0.65:
            # the head was centred on the 640x480 default scene), so the f16 noise -- relative to the MAGNITUDE -- is ~15 % of the spread
            # instead of the 2-3 % of tests/test_discriminative_gpu.py.  What is asserted: the two score vectors correlate, and the
            # hypothesis the library picked is near the top of the oracle's ranking of the same refined poses (teacher-forced regret)
            corr = float(np.corrcoef(scores, s_ref)[0, 1])
            rank = int((s_ref > s_ref[idx]).sum())
            regret = float((s_ref.max() - s_ref[idx]) / (s_ref.max() - np.median(s_ref)))
            print(f"    scores vs the oracle's (teacher-forced): corr {corr:.4f}, the library's winner {idx} has oracle rank {rank} (regret {regret:.3f}); "
                  f"score spread {s_ref.std():.3f} around {s_ref.mean():.2f}")
            assert idx == int(np.argmax(scores))
            assert corr > 0.95 and rank < 8 and regret < 0.3, (corr, rank, regret)
    finally:
        m.close()


@pytest.mark.parametrize("textured", [True, False])
def test_int8_720p_experimental_level_on_heldout_scenes(disc_nets, textured):
    """FP_PREC_INT8 is EXPERIMENTAL (it does not hold the configs[4] bar: the strict xfail below).  This is a REGRESSION GUARD of its
    measured level, not a parity claim: 1280x720, N = 252, calibrated on 16 frames of the scene family, measured on 6 OTHER frames
    against the f16 path of the same model.  Measured (round 6, profiles/r06_q8_blocks_*.log): share within 1 mm / 1 deg 85-90 % on
    average, 39-74 % on the worst scene, common mode 0.1-1.0 mm, de-meaned p95 0.8-1.6 mm, delta correlation > 0.99.  Guards: the run is
    deterministic (_heldout_stats), and the continuous quantities stay inside twice their measured range -- a broken kernel or a
    mis-applied record is off by orders of magnitude (round 4 calibrated on the measured frame: 1-12 mm on these scenes)."""
    rows = _heldout_table(disc_nets, 1280, 720, textured, FP_PREC_INT8, 6)
    frac = np.array([r["frac"] for r in rows]); cm = np.array([r["cm"] for r in rows])
    print(f"INT8 (experimental) 1280x720 textured={textured}: share within 1 mm / 1 deg mean {frac.mean() * 100:.1f} % (worst scene {frac.min() * 100:.1f} %, "
          f"{int((frac >= 0.95).sum())} of {len(rows)} scenes >= 95 %), common-mode mean {cm.mean():.2f} mm max {cm.max():.2f} mm")
    for r in rows:
        assert r["mm_p95"] < 3.0 and r["deg_p95"] < 1.0 and r["cm"] < 2.0, r
        assert r["corr"] > 0.95, r


@pytest.mark.xfail(strict=True, reason="INT8 does not hold the configs[4] bar on EVERY unseen scene (>= 95 % of the refined poses within 1 mm / 1 deg of f16 and a "
                                       "common-mode shift < 0.3 mm): 85-90 % on average, one or two of six scenes at 40-80 %, common mode up to 1 mm -- and no "
                                       "subset of trunk stages holds it either (tools/q8_blocks.py).  configs[4] ships in f16; INT8 is experimental.")
@pytest.mark.parametrize("textured", [True, False])
def test_int8_holds_95_percent_on_every_heldout_scene(disc_nets, textured):
    rows = _heldout_table(disc_nets, 1280, 720, textured, FP_PREC_INT8, 6)
    assert all(r["frac"] >= 0.95 and r["cm"] < 0.3 for r in rows), [(round(r["frac"], 3), round(r["cm"], 2)) for r in rows]


@pytest.mark.xfail(strict=True, reason="FP8 e4m3 does not hold the configs[4] bar (>= 95 % of the refined poses within 1 mm / 1 deg of f16): 3 mantissa bits on "
                                       "every activation are a per-element error of 2^-4 against a between-hypothesis signal of ~2 % of the feature scale -- "
                                       "measured 45-54 % even when calibrated on the measured frame (round 4).  The precision stays selectable and tested at "
                                       "kernel level and on Track; it is experimental, not a configs[4] path.")
def test_register_720p_fp8_meets_the_bar(disc_nets):
    rows = _heldout_table(disc_nets, 1280, 720, True, FP_PREC_FP8, 2)
    assert all(r["frac"] >= 0.95 for r in rows), [round(r["frac"], 3) for r in rows]


def test_stage_mask_of_the_8bit_trunk(disc_nets, syn_mesh, syn_scene):
    """[r6] run_trunk_q8 runs every residual stage either on 8-bit operands or on its 2-byte weights and kernels (Net::q8_blocks; test
    build: fpt_set_q8_blocks).  Invariants that do not need an error budget:
      * mask 0 -- an "INT8" network none of whose stages is 8-bit -- IS the f16 network: Track and a 42-hypothesis Register return the
        f16 path's poses bit for bit (same weights, same kernels, same buffers; run without the calibration's correction steps, whose
        arithmetic -- a mean of per-frame means against a pooled mean -- leaves ~1e-8 instead of 0);
      * every mask runs (all boundary forms: dual-output epilogues, q8_copy_kernel behind a 2-byte producer, 8-bit -> f16 only), is
        deterministic, and lands within the experimental precision's level of the f16 path;
      * a single 8-bit stage is closer to f16 than all four together."""
    def run(mask):
        # (a model of the TEST library: the product has no hook -- its mask is a compile-time constant)
        import subprocess, sys, json, textwrap
        code = textwrap.dedent(f"""
            import json, sys, numpy as np
            sys.path.insert(0, {ROOT!r})
            import torch
            from foundationpose_cpp_amd import FoundationPose, _lib, synthetic as syn
            from foundationpose_cpp_amd.api import FP_PREC_F16, FP_PREC_INT8
            _lib.use_test_lib()
            L = _lib.lib()
            L.fpt_set_q8_blocks({mask})
            if {mask} == 0:      # no correction steps: the calibration averages per-frame means where the 8-bit passes pool all frames, so its
                L.fpt_set_calib_opts(0, 0, 0)   # token / output corrections are ~1e-8 instead of 0 and flip an occasional f16 ulp of the positional table
            mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
            m = FoundationPose(mesh, syn.intrinsics(), {disc_nets[0]!r}, {disc_nets[1]!r})
            hyp = syn.perturb_pose(scene.gt_pose)
            held = syn.heldout_scenes(mesh, 1)[0]
            out = {{}}
            m.set_inplane_steps(1)
            for prec, name in ((FP_PREC_F16, "f16"), (FP_PREC_INT8, "int8")):
                if prec == FP_PREC_INT8:
                    m.calibrate_frames(syn.calibration_scenes(mesh, 4), mesh.name, prec)
                m.set_precision(prec)
                ok, t = m.Track(scene.rgb, scene.depth, hyp, mesh.name); assert ok, m.last_error
                ok, t2 = m.Track(scene.rgb, scene.depth, hyp, mesh.name); assert ok and np.array_equal(t, t2)
                ok, p, idx, sc, ref, _ = m.register_detailed(held.rgb, held.depth, held.mask, mesh.name); assert ok, m.last_error
                ok, p2, idx2, _, ref2, _ = m.register_detailed(held.rgb, held.depth, held.mask, mesh.name); assert ok and idx == idx2 and np.array_equal(ref, ref2)
                out[name] = dict(track=t.tolist(), refined=ref.tolist(), idx=int(idx))
            m.close()
            print("RESULT" + json.dumps(out))
        """)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1][6:])
        return {k: dict(track=np.array(v["track"], np.float32), refined=np.array(v["refined"], np.float32), idx=v["idx"]) for k, v in d.items()}

    res = {mask: run(mask) for mask in (0, 4, 6, 9, 15)}
    r0 = res[0]
    assert np.array_equal(r0["int8"]["track"], r0["f16"]["track"]) and np.array_equal(r0["int8"]["refined"], r0["f16"]["refined"]) and r0["int8"]["idx"] == r0["f16"]["idx"]
    cm = {}
    for mask, r in res.items():
        d = r["int8"]["refined"][:, :3, 3] - r["f16"]["refined"][:, :3, 3]
        cm[mask] = (float(np.linalg.norm(d, axis=1).max() * 1e3), float(_rot_deg(r["int8"]["refined"], r["f16"]["refined"]).max()))
        ang, dist = _pose_err(r["int8"]["track"], r["f16"]["track"])
        print(f"stage mask {mask:2d}: 42 refined poses vs f16 max {cm[mask][0]:.3f} mm / {cm[mask][1]:.3f} deg; Track vs f16 {dist * 1e3:.3f} mm / {ang:.3f} deg")
        # (regression guards of an experimental precision at twice the measured level -- measured: refined poses <= 2.2 mm / 0.31 deg, Track <= 0.91 mm / 0.27 deg)
        assert cm[mask][0] < 4.5 and cm[mask][1] < 1.0 and dist < 2e-3 and ang < 1.0, (mask, cm[mask], ang, dist)
    assert cm[0][0] == 0.0      # (bit-identical above; the ANGLE of R R^T evaluated in float32 is sqrt(2 eps) ~ 0.03 deg for identical matrices, not 0)
    assert cm[4][0] < cm[15][0], cm


def test_int8_refiner_rows_follow_the_oracle_on_a_heldout_scene(disc_nets, syn_mesh):
    """The ORACLE leg under the discriminating weights (round-4 review: the 8-bit paths were only ever compared with the HIP f16 path):
    refiner outputs of the INT8 network for 42 hypotheses of a held-out scene against the torch fp32 network on the same crops --
    de-meaned against the between-hypothesis spread like tests/test_discriminative_gpu.py::test_refiner_rows_follow_torch (f16: rms <= 2 %,
    worst row <= 5 %, common mode <= 10 % of the spread).  INT8, measured: rms 8-11 %, worst row 25-35 %, common mode 5-25 %."""
    blob = _calibrated_int8_blob(disc_nets, 640, 480, True)
    scene = syn.heldout_scenes(syn_mesh, 2)[1]
    m = FoundationPose(syn_mesh, syn.intrinsics(), disc_nets[0], disc_nets[1])
    try:
        m.set_calibration_blob(blob)
        m.upload_frame(scene.rgb, scene.depth)
        poses = m.get_hyp_poses(scene.mask)[::6]
        poses = np.stack([syn.perturb_pose(p, deg=3.0, trans=0.006, seed=100 + i) for i, p in enumerate(poses)])   # own observed crop per hypothesis
        a, b = m.render_and_transform(syn_mesh.name, poses, 1.2)
        with torch.no_grad():
            rt, rr = (o.numpy() for o in disc_nets[2](torch.from_numpy(a), torch.from_numpy(b)))
        out = {}
        for prec, name in ((FP_PREC_F16, "f16"), (FP_PREC_INT8, "int8")):
            m.set_precision(prec)
            out[name] = m.refiner_infer(a, b)
        for name, rms_bar, max_bar, cm_bar in (("f16", 0.03, 0.08, 0.10), ("int8", 0.15, 0.45, 0.35)):
            for got, ref, what in ((out[name][0], rt, "trans"), (out[name][1], rr, "rot")):
                spread = ref.std(0)
                err = (got - got.mean(0)) - (ref - ref.mean(0))
                rms, mx, cmv = np.sqrt((err ** 2).mean(0)) / spread, np.abs(err).max(0) / spread, np.abs(got.mean(0) - ref.mean(0)) / spread
                print(f"{name} {what} vs torch fp32 on a held-out scene: de-meaned rms {rms.max() * 100:.1f} %, worst row {mx.max() * 100:.1f} %, common mode {cmv.max() * 100:.1f} % of the spread")
                assert rms.max() <= rms_bar and mx.max() <= max_bar and cmv.max() <= cm_bar, (name, what, rms, mx, cmv)
    finally:
        m.close()


def test_int8_picks_the_f16_winner_when_there_is_a_clear_one(disc_nets, syn_mesh):
    """Winner parity needs a fixture that can show it (round-4 review).  Two things blur it end to end: with 252 hypotheses several
    converge on one pose and score within a hair of each other, and rendering is discontinuous in the pose -- the 0.5 mm by which INT8
    and f16 refined poses differ moves a score by more than the top-2 gap (first version of this test: f16 gap 27 %, the two pipelines
    picked hypotheses 122 deg apart, each the best of ITS refined poses: teacher-forced regret 0).  So the fixture fixes the inputs: the
    f16 path's refined poses of 42 distinct views (one in-plane step) are rendered once and BOTH scorers score the same crops; a scene
    qualifies when the f16 top-2 gap is >= 30 % of (best - median) -- the INT8 scorer's per-hypothesis error is ~9 % rms / ~25 % worst row
    of the spread (oracle leg above), so a 25 % gap can still flip (measured: 1 of 3 such scenes), and a 34 % gap did once -- and there the
    INT8 scorer must pick the f16 winner or, at most once in four such scenes, its runner-up (never with a gap >= 50 %); over ALL 24
    scenes the two scorers agree on >= 75 %.  End to end the statement is the teacher-forced one of
    test_register_720p_int8_on_heldout_scenes (rank 0, regret 0)."""
    blob = _calibrated_int8_blob(disc_nets, 640, 480, True)
    m = FoundationPose(syn_mesh, syn.intrinsics(), disc_nets[0], disc_nets[1])
    checked = agree = clear_equal = 0
    n_scenes = 24
    try:
        m.set_calibration_blob(blob)
        m.set_inplane_steps(1)
        for k, scene in enumerate(syn.heldout_scenes(syn_mesh, n_scenes)):
            m.set_precision(FP_PREC_F16)
            ok, _, _, _, ref16, _ = m.register_detailed(scene.rgb, scene.depth, scene.mask, syn_mesh.name)
            assert ok, m.last_error
            m.upload_frame(scene.rgb, scene.depth)
            a, b = m.render_and_transform(syn_mesh.name, ref16, 1.1)
            sc16 = m.scorer_infer(a, b)
            o = np.sort(sc16)[::-1]
            gap = (o[0] - o[1]) / (o[0] - np.median(sc16))
            m.set_precision(FP_PREC_INT8)
            sc8 = m.scorer_infer(a, b)
            print(f"held-out scene {k}: f16 top-2 gap {gap * 100:.0f} % of (best - median): winner f16 {int(sc16.argmax())} / int8 {int(sc8.argmax())}, "
                  f"score corr {np.corrcoef(sc8, sc16)[0, 1]:.4f}")
            agree += int(sc8.argmax()) == int(sc16.argmax())
            if gap < 0.30:
                continue
            checked += 1
            rank16 = int((sc16 > sc16[int(sc8.argmax())]).sum())     # where the f16 scorer ranks the INT8 scorer's winner
            clear_equal += rank16 == 0
            # a clear f16 winner: INT8 picks it, or at worst the f16 runner-up (per-row error ~9 % rms / ~25 % worst row of the spread: a gap of
            # 30-40 % flips now and then -- one of the four such scenes in the round's last run); a gap of half the spread never flips
            assert rank16 <= 1, (k, int(sc8.argmax()), int(sc16.argmax()), gap)
            assert rank16 == 0 or gap < 0.50, (k, int(sc8.argmax()), int(sc16.argmax()), gap)
        print(f"same winner on {agree} of {n_scenes} scenes; {checked} scenes with a clear f16 winner (gap >= 30 %), {clear_equal} of them equal")
        assert checked >= 3, f"only {checked} of {n_scenes} held-out scenes have a clear f16 winner: the fixture does not discriminate"
        assert clear_equal >= checked - max(1, checked // 4), (clear_equal, checked)
        assert agree >= 0.75 * n_scenes, agree
    finally:
        m.close()


def test_calibration_session_api(disc_nets, syn_mesh, syn_scene):
    """fp_calibrate_begin / _add_frame / _finish / _abort: argument and state errors, a failed calibration leaves the previous record in
    place (or the precision uncalibrated), a multi-frame calibration is reproducible byte for byte and survives the blob round trip."""
    L = _lib.lib()
    scenes = syn.calibration_scenes(syn_mesh, 3)
    m = FoundationPose(syn_mesh, syn.intrinsics(), disc_nets[0], disc_nets[1])
    m2 = FoundationPose(syn_mesh, syn.intrinsics(), disc_nets[0], disc_nets[1])
    try:
        assert L.fp_calibrate_frames(m.handle) == -1
        assert L.fp_calibrate_finish(m.handle) != 0 and "fp_calibrate_begin" in _lib.last_error()
        assert L.fp_calibrate_begin(m.handle, FP_PREC_F16) != 0
        assert L.fp_calibrate_begin(m.handle, FP_PREC_INT8) == 0 and L.fp_calibrate_frames(m.handle) == 0
        assert L.fp_calibrate_finish(m.handle) != 0 and "no calibration frame" in _lib.last_error()      # (finishing closes the session)
        assert L.fp_calibrate_frames(m.handle) == -1
        with pytest.raises(Exception):
            m.set_precision(FP_PREC_INT8)                     # still uncalibrated
        # a frame whose mask is empty makes the calibration fail: no record appears
        bad = [(scenes[0].rgb, scenes[0].depth, scenes[0].mask), (scenes[1].rgb, scenes[1].depth, np.zeros_like(scenes[1].mask))]
        with pytest.raises(Exception) as e:
            m.calibrate_frames(bad, syn_mesh.name, FP_PREC_INT8)
        assert "Mask is all zero" in str(e.value)
        with pytest.raises(Exception):
            m.get_calibration_blob(FP_PREC_INT8)
        ok, _ = m.Track(syn_scene.rgb, syn_scene.depth, syn.perturb_pose(syn_scene.gt_pose), syn_mesh.name)    # the model is still usable
        assert ok, m.last_error
        # a good one, twice: identical records; then a FAILED re-calibration keeps the good record
        m.calibrate_frames(scenes, syn_mesh.name, FP_PREC_INT8)
        blob = m.get_calibration_blob(FP_PREC_INT8)
        m2.calibrate_frames(scenes, syn_mesh.name, FP_PREC_INT8)
        assert m2.get_calibration_blob(FP_PREC_INT8) == blob
        with pytest.raises(Exception):
            m.calibrate_frames(bad, syn_mesh.name, FP_PREC_INT8)
        assert m.get_calibration_blob(FP_PREC_INT8) == blob
        m.set_precision(FP_PREC_INT8)
        hyp = syn.perturb_pose(syn_scene.gt_pose)
        ok, p_a = m.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
        assert ok, m.last_error
        m3 = FoundationPose(syn_mesh, syn.intrinsics(), disc_nets[0], disc_nets[1])
        try:
            m3.set_calibration_blob(blob)
            m3.set_precision(FP_PREC_INT8)
            ok, p_b = m3.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
            assert ok and np.array_equal(p_a, p_b)            # the record reproduces the networks bit for bit (weights rounded against its frame means)
        finally:
            m3.close()
    finally:
        m.close()
        m2.close()


def test_calibration_is_reproducible(disc_nets, syn_mesh, syn_scene):
    """Two models calibrated on the same frame hold the same blob, byte for byte: the per-channel statistics are combined with integer
    atomics (chan_stats_kernel), everything downstream of them is deterministic."""
    blobs = []
    for _ in range(2):
        m = FoundationPose(syn_mesh, syn.intrinsics(), disc_nets[0], disc_nets[1])
        try:
            m.calibrate(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name, FP_PREC_INT8)
            blobs.append(m.get_calibration_blob(FP_PREC_INT8))
        finally:
            m.close()
    assert blobs[0] == blobs[1]


def test_q8_per_layer_error_vs_fp32(nets, syn_mesh, syn_scene):
    """Error budget of the 8-bit trunks under the PLAIN weights: refiner and scorer outputs on 16 hypotheses against the torch fp32
    networks.  f16 lands at ~1e-3 of the output scale; the 8-bit types are asserted at 5e-2 (FP8) / 2e-2 (INT8)."""
    m = FoundationPose(syn_mesh, syn.intrinsics(), nets[0], nets[1])
    try:
        m.upload_frame(syn_scene.rgb, syn_scene.depth)
        poses = m.get_hyp_poses(syn_scene.mask)[:16]
        a, b = m.render_and_transform(syn_mesh.name, poses, 1.2)
        with torch.no_grad():
            rt, rr = nets[2](torch.from_numpy(a), torch.from_numpy(b))
            rs = nets[3](torch.from_numpy(a), torch.from_numpy(b)).numpy()
        scale_t, scale_r, scale_s = np.abs(rt.numpy()).max(), np.abs(rr.numpy()).max(), np.abs(rs).max()
        e = {}
        for prec, name in ((FP_PREC_F16, "f16"), (FP_PREC_FP8, "fp8"), (FP_PREC_INT8, "int8")):
            if prec != FP_PREC_F16:
                m.set_precision(FP_PREC_F16)
                m.calibrate(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name, prec)
            m.set_precision(prec)
            t, r = m.refiner_infer(a, b)
            s = m.scorer_infer(a, b)
            e[f"trans {name}"] = np.abs(t - rt.numpy()).max() / scale_t
            e[f"rot {name}"] = np.abs(r - rr.numpy()).max() / scale_r
            e[f"score {name}"] = np.abs(s - rs).max() / scale_s
        print("max |err| / max |output| vs torch fp32:", {k: float(f"{v:.3e}") for k, v in e.items()})
        for k, v in e.items():
            assert v < (5e-2 if "fp8" in k else 2e-2 if "int8" in k else 1e-2), (k, v)
    finally:
        m.close()
