"""Parity of the MFMA networks against the PyTorch fp32 reference (oracle/nets_torch.py) with identical weights.

Floating point: the HIP path stores activations/weights in fp16 and accumulates in fp32 (the reference's TensorRT
engines are fp16 too, tools/cvt_onnx2trt.bash:3-15 `--fp16`).  Tolerances are written at each assert.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from foundationpose_cpp_amd import FoundationPose, _lib, synthetic as syn, weights as W
from oracle import fp_oracle as fo
from oracle import nets_torch as NT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _h(a):  # fp16 round trip (what the device stores)
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def _conv_hip(x, w_oihw, bias, stride, pad, relu, res=None, split=0):
    L = _lib.test_lib()   # kernel-level hooks live in the test build (same sources + -DFP_TEST_HOOKS)
    NB, H, Wd, Cin = x.shape
    Cout, _, KH, KW = w_oihw.shape
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (Wd + 2 * pad - KW) // stride + 1
    if KH == 4 and pad == 2 and stride == 1:      # the kernel's space-to-depth stem mode (pad 2 before, 1 after)
        OH, OW = H, Wd
    wk = np.ascontiguousarray(w_oihw.transpose(0, 2, 3, 1), np.float32)
    out = np.zeros((NB, OH, OW, Cout) if split == 0 else (NB - split, OH, OW, 2 * Cout), np.float32)
    x = np.ascontiguousarray(x, np.float32)
    b = np.ascontiguousarray(bias, np.float32)
    r = np.ascontiguousarray(res, np.float32) if res is not None else None
    rc = L.fpt_conv(_p(x), _p(wk), _p(b), _p(r) if r is not None else None, NB, H, Wd, Cin, Cout, KH, KW, stride, pad,
                    OH, OW, int(relu), split, _p(out), 1, None)
    assert rc == 0, _lib.last_error()
    return out


def _conv_ref(x, w, bias, stride, pad, relu, res=None):
    y = torch.nn.functional.conv2d(torch.from_numpy(_h(x)).permute(0, 3, 1, 2), torch.from_numpy(_h(w)),
                                   torch.from_numpy(np.asarray(bias, np.float32)), stride, pad).permute(0, 2, 3, 1)
    if res is not None:
        y = y + torch.from_numpy(_h(res))
    if relu:
        y = torch.relu(y)
    return y.numpy()


@pytest.mark.parametrize("shape", [
    # NB, H, W, Cin, Cout, k, stride, relu, use_res
    (3, 40, 40, 128, 128, 3, 1, True, True),      # encodeA res block; M = 4800 (not a multiple of 128)
    (2, 80, 80, 64, 128, 3, 2, True, False),      # encodeA.1 stride 2
    (2, 40, 40, 256, 512, 3, 2, True, False),     # encodeAB.2
    (1, 20, 20, 512, 512, 3, 1, False, True),     # encodeAB 512 block
    (1, 400, 1, 512, 1536, 1, 1, False, False),   # QKV projection as 1x1 conv
    (1, 7, 1, 512, 512, 1, 1, False, False),      # tiny M (cross attention over 7 hypotheses)
    (2, 9, 11, 64, 64, 3, 1, True, False),        # BN=64 tile path, ragged image
])
def test_conv_igemm_matches_torch(shape):
    NB, H, Wd, Cin, Cout, k, stride, relu, use_res = shape
    rng = np.random.default_rng(1)
    x = rng.normal(size=(NB, H, Wd, Cin)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    pad = (k - 1) // 2
    OH = (H + 2 * pad - k) // stride + 1
    OW = (Wd + 2 * pad - k) // stride + 1
    res = rng.normal(size=(NB, OH, OW, Cout)).astype(np.float32) if use_res else None
    got = _conv_hip(x, w, b, stride, pad, relu, res)
    ref = _conv_ref(x, w, b, stride, pad, relu, res)
    # fp32 accumulate of fp16 products, output rounded to fp16: |err| <= 2^-11 |y| + accumulation-order noise
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("variant", [3, 5, 7, 8])
def test_conv_alternative_schedules_match_torch(variant):
    """every schedule a layer can reach: 256x128 ping-pong everywhere (3), 256x256 rounds without the resident-halo
    kernels (5), resident-halo kernels forced (7) / disabled (8)"""
    L = _lib.test_lib()
    rng = np.random.default_rng(5)
    try:
        L.fpt_set_conv_variant(variant)
        for (NB, H, Cin, Cout, k, stride, use_res) in [(9, 40, 128, 128, 3, 1, True), (5, 40, 256, 256, 3, 1, False),
                                                       (70, 20, 512, 512, 3, 1, True), (90, 40, 256, 256, 3, 1, True), (170, 40, 128, 128, 3, 1, True), (3, 80, 64, 128, 3, 2, False),
                                                       (40, 400, 512, 1536, 1, 1, False), (5, 80, 32, 64, 4, 1, False)]:
            W_ = 1 if k == 1 else H
            x = rng.normal(size=(NB, H, W_, Cin)).astype(np.float32)
            w = (rng.normal(size=(Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
            b = rng.normal(size=Cout).astype(np.float32)
            pad = 2 if k == 4 else (k - 1) // 2
            OH = H if k == 4 else (H + 2 * pad - k) // stride + 1
            OW = W_ if k == 4 else (W_ + 2 * pad - k) // stride + 1
            res = rng.normal(size=(NB, OH, OW, Cout)).astype(np.float32) if use_res else None
            got = _conv_hip(x, w, b, stride, pad, True, res)
            if k == 4:    # asymmetric-padding stem mode: reference = pad (2,1)
                xt = torch.nn.functional.pad(torch.from_numpy(_h(x)).permute(0, 3, 1, 2), (2, 1, 2, 1))
                ref = torch.relu(torch.nn.functional.conv2d(xt, torch.from_numpy(_h(w)), torch.from_numpy(b))).permute(0, 2, 3, 1).numpy()
            else:
                ref = _conv_ref(x, w, b, stride, pad, True, res)
            np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3)
    finally:
        L.fpt_set_conv_variant(0)


@pytest.mark.parametrize("shape", [
    # NB, H, W, Cin, Cout, k, stride, use_res        128-byte K-steps, tile
    (1, 80, 80, 32, 64, 4, 1, False),      # stem: 8 steps, 32-channel tiles (the Cout = 64 row permutation)
    (1, 80, 80, 64, 128, 3, 2, False),     # encodeA.1: 9 steps (the waves get 3 / 2 / 2 / 2)
    (1, 40, 40, 128, 128, 3, 1, True),     # 18 steps
    (1, 40, 40, 256, 256, 3, 1, True),     # 36 steps
    (1, 40, 40, 256, 512, 3, 2, False),    # encodeAB.2: 36 steps, 16-pixel tiles
    (1, 20, 20, 512, 512, 3, 1, True),     # 72 steps
    (1, 13, 7, 128, 128, 3, 1, True),      # M = 91: the last tile has 11 real pixels
    (2, 400, 1, 512, 512, 1, 1, True),     # Linear layer, 8 steps
    (1, 3, 1, 512, 1536, 1, 1, False),     # M = 3
])
def test_small_problem_kernel_matches_torch_and_its_first_version(shape):
    """conv_smallx_kernel (weights in fragment order, pixels through a per-wave LDS-DMA ring, hand-counted vmcnt) against torch,
    and BIT-identical to conv_smallm_kernel (input pixels global -> registers in operand shape, compiler-counted waits): same tiles,
    same K split over the waves, same reduction order -- only the way the operands travel differs."""
    NB, H, Wd, Cin, Cout, k, stride, use_res = shape
    L = _lib.test_lib()
    rng = np.random.default_rng(11)
    x = rng.normal(size=(NB, H, Wd, Cin)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    pad = 2 if k == 4 else (k - 1) // 2
    OH = H if k == 4 else (H + 2 * pad - k) // stride + 1
    OW = Wd if k == 4 else (Wd + 2 * pad - k) // stride + 1
    res = rng.normal(size=(NB, OH, OW, Cout)).astype(np.float32) if use_res else None
    if k == 4:
        xt = torch.nn.functional.pad(torch.from_numpy(_h(x)).permute(0, 3, 1, 2), (2, 1, 2, 1))
        ref = torch.relu(torch.nn.functional.conv2d(xt, torch.from_numpy(_h(w)), torch.from_numpy(b))).permute(0, 2, 3, 1).numpy()
    else:
        ref = _conv_ref(x, w, b, stride, pad, True, res)
    try:
        L.fpt_set_smallm(2)       # small-problem kernel whatever the K length
        got = _conv_hip(x, w, b, stride, pad, True, res)
        L.fpt_set_smallm(3)       # its first version
        first = _conv_hip(x, w, b, stride, pad, True, res)
        L.fpt_set_smallm(0)       # the LDS-ring / split-K schedules
        other = _conv_hip(x, w, b, stride, pad, True, res)
    finally:
        L.fpt_set_smallm(1)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3)
    np.testing.assert_array_equal(got, first)
    np.testing.assert_allclose(got, other, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("rows,Cout,relu,use_res", [(66001, 512, True, True), (11111, 1536, False, False)])
def test_linear_layers_two_workgroups_per_cu_kernel(rows, Cout, relu, use_res):
    """gemm_k32_kernel (Linear layers with >= 512 tiles): ragged last tile, residual + ReLU epilogue, both n-tile counts"""
    rng = np.random.default_rng(11)
    x = rng.normal(size=(rows, 1, 1, 512)).astype(np.float32)
    w = (rng.normal(size=(Cout, 512, 1, 1)) / np.sqrt(512)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    res = rng.normal(size=(rows, 1, 1, Cout)).astype(np.float32) if use_res else None
    got = _conv_hip(x, w, b, 1, 0, relu, res)
    ref = _h(x).reshape(rows, 512) @ _h(w).reshape(Cout, 512).T + b
    if use_res:
        ref = ref + _h(res).reshape(rows, Cout)
    if relu:
        ref = np.maximum(ref, 0)
    np.testing.assert_allclose(got.reshape(rows, Cout), ref, rtol=2e-3, atol=2e-3)
    # asymmetric operands: a transposed fragment or a wrong channel permutation cannot pass
    x2 = np.zeros((rows, 1, 1, 512), np.float32)
    x2[:, 0, 0, :] = (np.arange(rows)[:, None] % 97) / 97.0 + np.arange(512)[None, :] / 512.0
    w2 = np.zeros((Cout, 512, 1, 1), np.float32)
    w2[np.arange(Cout), (np.arange(Cout) * 5) % 512, 0, 0] = 1.0 + np.arange(Cout) / Cout
    got2 = _conv_hip(x2, w2, np.zeros(Cout, np.float32), 1, 0, False, None)
    ref2 = _h(x2).reshape(rows, 512) @ _h(w2).reshape(Cout, 512).T
    np.testing.assert_allclose(got2.reshape(rows, Cout), ref2, rtol=2e-3, atol=2e-3)


def test_conv_transpose_detecting_and_split_store():
    # asymmetric weights/inputs (a transposed fragment layout cannot pass) + the a|b channel-concat epilogue
    NB, H, Cin, Cout = 4, 8, 128, 128
    x = np.zeros((NB, H, H, Cin), np.float32)
    x[..., :] = np.arange(Cin)[None, None, None, :] / Cin
    x += np.arange(NB)[:, None, None, None] * 0.5 + np.arange(H)[None, :, None, None] * 0.01
    w = np.zeros((Cout, Cin, 3, 3), np.float32)
    for co in range(Cout):
        w[co, (co * 7) % Cin, co % 3, (co // 3) % 3] = 1.0 + co / Cout
    b = np.linspace(-1, 1, Cout).astype(np.float32)
    got = _conv_hip(x, w, b, 1, 1, False, None, split=2)
    ref = _conv_ref(x, w, b, 1, 1, False)
    ref_cat = np.concatenate([ref[:2], ref[2:]], axis=-1)      # torch.cat((x[:bs], x[bs:]), 1)
    np.testing.assert_allclose(got, ref_cat, rtol=2e-3, atol=2e-3)


def test_stem_space_to_depth_equals_7x7_stride2():
    # the 7x7/s2/p3 stem is evaluated as a 4x4/s1 conv over the space-to-depth input (fp_nn.hip make_stem)
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, size=(2, 160, 160, 6)).astype(np.float32)
    w = (rng.normal(size=(64, 6, 7, 7)) / np.sqrt(6 * 49)).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    ref = _conv_ref(x, w, b, 2, 3, True)
    x8 = np.concatenate([x, np.zeros((2, 160, 160, 2), np.float32)], -1)
    s2d = x8.reshape(2, 80, 2, 80, 2, 8).transpose(0, 1, 3, 2, 4, 5).reshape(2, 80, 80, 32)
    ws = np.zeros((64, 32, 4, 4), np.float32)
    for a in range(4):
        for bb in range(4):
            for dy in range(2):
                for dx in range(2):
                    kh, kw = 2 * a + dy - 1, 2 * bb + dx - 1
                    if kh >= 0 and kw >= 0:
                        ws[:, (dy * 2 + dx) * 8:(dy * 2 + dx) * 8 + 6, a, bb] = w[:, :, kh, kw]
    got = _conv_hip(s2d, ws, b, 1, 2, True)          # pad 2 + KH 4 + stride 1 -> the kernel's s2d stem mode (80x80 out)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("B,T", [(2, 400), (1, 252), (1, 7), (3, 33), (1, 1), (1, 2016), (2, 129)])
def test_attention_matches_torch(B, T):
    rng = np.random.default_rng(3)
    qkv = (rng.normal(size=(B, T, 1536)) * 1.5).astype(np.float32)
    qkv[0, T // 2, :512] *= 4            # a spiked query row exercises the online-softmax rescale
    out = np.zeros((B, T, 512), np.float32)
    assert _lib.test_lib().fpt_attention(_p(qkv), B, T, _p(out)) == 0, _lib.test_lib().fp_last_error()
    q, k, v = [torch.from_numpy(_h(qkv[..., i * 512:(i + 1) * 512])).reshape(B, T, 4, 128).permute(0, 2, 1, 3) for i in range(3)]
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(B, T, 512).numpy()
    np.testing.assert_allclose(out, ref, rtol=5e-3, atol=5e-3)   # P is rounded to fp16 before the PV MFMA


@pytest.fixture(scope="module")
def nets(tmp_path_factory):
    d = tmp_path_factory.mktemp("w")
    rp, sp = str(d / "refiner.fpw"), str(d / "scorer.fpw")
    rs = W.pack_synthetic("refiner", rp)
    ss = W.pack_synthetic("scorer", sp)
    return rp, sp, NT.build("refiner", rs), NT.build("scorer", ss)


@pytest.fixture(scope="module")
def model(nets, syn_mesh):
    m = FoundationPose(syn_mesh, syn.intrinsics(), nets[0], nets[1])
    yield m
    m.close()


def _blobs(model, syn_mesh, syn_scene, n, crop_ratio):
    model.upload_frame(syn_scene.rgb, syn_scene.depth)
    poses = model.get_hyp_poses(syn_scene.mask)[:n]
    a, b = model.render_and_transform(syn_mesh.name, poses, crop_ratio)
    return poses, a, b


def test_refiner_matches_torch(model, nets, syn_mesh, syn_scene):
    poses, a, b = _blobs(model, syn_mesh, syn_scene, 6, 1.2)
    trans, rot = model.refiner_infer(a, b)
    with torch.no_grad():
        rt, rr = nets[2](torch.from_numpy(a), torch.from_numpy(b))
    # 13 fp16 conv layers + transformer: relative error budget ~1e-2 of the output scale (|trans|,|rot| ~ 0.1)
    np.testing.assert_allclose(trans, rt.numpy(), rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(rot, rr.numpy(), rtol=2e-2, atol=2e-3)


def test_scorer_matches_torch(model, nets, syn_mesh, syn_scene):
    poses, a, b = _blobs(model, syn_mesh, syn_scene, 9, 1.1)
    scores = model.scorer_infer(a, b)
    with torch.no_grad():
        ref = nets[3](torch.from_numpy(a), torch.from_numpy(b)).numpy()
    np.testing.assert_allclose(scores, ref, rtol=2e-2, atol=3e-3)


def test_product_library_schedules_by_batch_size(model, nets, syn_mesh, syn_scene):
    """Round-3 review, weak #11: the kernel-level schedule tests run on the TEST build (fpt_* hooks); this one drives the PRODUCT
    library (`model` = libfoundationpose_amd.so through the C ABI) at the batch sizes where run_conv_dt changes schedule layer by layer
    -- 1 (Track: conv_smallx_kernel, grouped heads, split-key attention), 3 and 12 (small-problem / mid-sized implicit-GEMM paths), 33
    (the resident-halo kernels switch on), 64 and 130 (256x256 ping-pong rounds + ping-pong cascade + deep-ring left-overs) -- against
    the torch fp32 networks: what ships is what is tested.  One set of crops, prefixes of it (the networks are per-hypothesis)."""
    _, a, b = _blobs(model, syn_mesh, syn_scene, 130, 1.2)
    with torch.no_grad():
        rt, rr = nets[2](torch.from_numpy(a), torch.from_numpy(b))
        feats = nets[3].extract_feat(torch.from_numpy(a), torch.from_numpy(b))
    rt, rr = rt.numpy(), rr.numpy()
    for n in (1, 3, 12, 33, 64, 130):
        t, r = model.refiner_infer(a[:n], b[:n])
        np.testing.assert_allclose(t, rt[:n], rtol=2e-2, atol=2e-3, err_msg=f"refiner trans, batch {n}")
        np.testing.assert_allclose(r, rr[:n], rtol=2e-2, atol=2e-3, err_msg=f"refiner rot, batch {n}")
        s = model.scorer_infer(a[:n], b[:n])
        with torch.no_grad():
            ref = nets[3].head(feats[:n]).numpy()      # (the cross-hypothesis attention sees the n hypotheses of the batch)
        np.testing.assert_allclose(s, ref, rtol=2e-2, atol=3e-3, err_msg=f"scores, batch {n}")


_BUILD_EQ_CODE = """
import sys, numpy as np
sys.path.insert(0, {root!r})
import torch
from foundationpose_cpp_amd import FoundationPose, _lib, synthetic as syn
from foundationpose_cpp_amd.api import FP_PREC_F16, FP_PREC_BF16, FP_PREC_INT8
if {test_build}:
    _lib.use_test_lib()
mesh = syn.make_mesh(); scene = syn.make_scene(mesh)
m = FoundationPose(mesh, syn.intrinsics(), {rp!r}, {sp!r})
out = {{}}
m.upload_frame(scene.rgb, scene.depth)
poses = m.get_hyp_poses(scene.mask)[:130]
a, b = m.render_and_transform(mesh.name, poses, 1.2)
out["render"], out["crop"] = a, b
for n in (1, 12, 33, 130):
    t, r = m.refiner_infer(a[:n], b[:n])
    out[f"trans{{n}}"], out[f"rot{{n}}"], out[f"score{{n}}"] = t, r, m.scorer_infer(a[:n], b[:n])
hyp = syn.perturb_pose(scene.gt_pose)
for prec, name in ((FP_PREC_F16, "f16"), (FP_PREC_BF16, "bf16"), (FP_PREC_INT8, "int8")):
    if prec == FP_PREC_INT8:
        m.set_precision(FP_PREC_F16)
        m.calibrate_frames(syn.calibration_scenes(mesh, 2), mesh.name, prec)
    m.set_precision(prec)
    for k in range(2):      # eager, then the captured graph
        ok, p = m.Track(scene.rgb, scene.depth, hyp, mesh.name); assert ok, m.last_error
        out[f"track_{{name}}_{{k}}"] = p
    m.set_inplane_steps(1)
    ok, p, idx, sc, ref, feat = m.register_detailed(scene.rgb, scene.depth, scene.mask, mesh.name); assert ok, m.last_error
    out[f"reg_{{name}}_scores"], out[f"reg_{{name}}_refined"], out[f"reg_{{name}}_feat"], out[f"reg_{{name}}_idx"] = sc, ref, feat, np.array([idx])
    m.set_inplane_steps(6)
m.set_precision(FP_PREC_F16)
ok, p = m.Register(scene.rgb, scene.depth, scene.mask, mesh.name); assert ok, m.last_error
out["register252"] = p
m.close()
np.savez({out!r}, **out)
"""


def test_product_and_test_builds_compute_the_same_bits(nets, tmp_path):
    """Round-5 review, weak #13: the kernel-level parity tests call `fpt_*` hooks of libfoundationpose_amd_test.so, "a different code object
    from the product".  The two libraries are the same sources (the test build adds hooks whose defaults are the product's compile-time
    constants), so with no hook touched they must compute THE SAME BITS: rendered and observed crops, both networks at the batch sizes where
    the schedules change (1, 12, 33, 130), Track (eager call and graph replay) and a 42-hypothesis Register in f16, bf16 and INT8, and the
    252-hypothesis Register -- every array of the product, bit for bit, equals the test build's.  What the kernel-level tests establish on
    the test build therefore holds for the code object that ships."""
    import subprocess
    import sys
    res = {}
    for build, flag in (("product", False), ("test", True)):
        out = str(tmp_path / f"{build}.npz")
        code = _BUILD_EQ_CODE.format(root=ROOT, test_build=flag, rp=nets[0], sp=nets[1], out=out)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (build, r.stderr[-3000:])
        with np.load(out) as z:
            res[build] = {k: z[k] for k in z.files}
    assert set(res["product"]) == set(res["test"]) and len(res["product"]) >= 30
    diff = [k for k in res["product"] if not np.array_equal(res["product"][k], res["test"][k])]
    assert not diff, diff


def _oracle_register(nets, mesh, scene, n_hyp):
    """Register restated with the oracle geometry + torch networks (foundationpose.cpp:181-228)."""
    om = fo.OracleMesh(mesh)
    poses = fo.get_hyp_poses(scene.depth, scene.mask, scene.K)[:n_hyp]
    a = fo.render(om, poses, scene.K, scene.depth.shape, 1.2)
    b = fo.crop(scene.rgb, scene.depth, scene.K, poses, 1.2, mesh.diameter)
    with torch.no_grad():
        t, r = nets[2](torch.from_numpy(a), torch.from_numpy(b))
    refined = fo.refine_post_process(poses, t.numpy(), r.numpy(), mesh.diameter)
    a = fo.render(om, refined, scene.K, scene.depth.shape, 1.1)
    b = fo.crop(scene.rgb, scene.depth, scene.K, refined, 1.1, mesh.diameter)
    with torch.no_grad():
        s = nets[3](torch.from_numpy(a), torch.from_numpy(b)).numpy()
    return syn.from_colmajor(refined), s


def _pose_err(a, b):
    dR = a[:3, :3] @ b[:3, :3].T
    ang = np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
    return ang, np.linalg.norm(a[:3, 3] - b[:3, 3])


def test_track_end_to_end(model, nets, syn_mesh, syn_scene):
    hyp = syn.perturb_pose(syn_scene.gt_pose)
    ok, pose = model.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
    assert ok, model.last_error
    om = fo.OracleMesh(syn_mesh)
    p16 = syn.to_colmajor(hyp[None])
    a = fo.render(om, p16, syn_scene.K, (480, 640), 1.2)
    b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, p16, 1.2, syn_mesh.diameter)
    with torch.no_grad():
        t, r = nets[2](torch.from_numpy(a), torch.from_numpy(b))
    ref = syn.from_colmajor(fo.refine_post_process(p16, t.numpy(), r.numpy(), syn_mesh.diameter))[0]
    ang, dist = _pose_err(pose, ref)
    assert ang < 0.1 and dist < 1e-4, (ang, dist)      # north_star bar is 1 deg / 1 mm; fp16 path lands far inside
    # reference error behaviour: unknown target -> False
    ok, _ = model.Track(syn_scene.rgb, syn_scene.depth, hyp, "nope")
    assert not ok and "target_name" in model.last_error
    # Track from a host frame uploads only the rows of the observed-crop window: a model that has never seen the whole frame (its
    # copy is otherwise uninitialised) must return the same pose, and the stage operators must refuse the partial frame
    m2 = FoundationPose(syn_mesh, syn.intrinsics(), nets[0], nets[1])
    try:
        ok, pose2 = m2.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
        assert ok and np.array_equal(pose2, pose)
        assert m2.get_hyp_poses(syn_scene.mask) is None and "crop window" in m2.last_error
        m2.upload_frame(syn_scene.rgb, syn_scene.depth)
        assert len(m2.get_hyp_poses(syn_scene.mask)) == 252
    finally:
        m2.close()
    # zero refine iterations: the reference's loop does not run and the hypothesis comes back unchanged
    ok, same = model.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name, refine_itr=0)
    assert ok and np.array_equal(same, hyp.astype(np.float32))
    # two iterations == two single-iteration calls chained (the pose travels through host-pinned memory in both directions)
    ok, p1 = model.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name)
    ok2, p2 = model.Track(syn_scene.rgb, syn_scene.depth, p1, syn_mesh.name)
    ok3, p12 = model.Track(syn_scene.rgb, syn_scene.depth, hyp, syn_mesh.name, refine_itr=2)
    assert ok and ok2 and ok3
    np.testing.assert_allclose(p12, p2, atol=1e-6)


@pytest.mark.parametrize("case", range(6))
def test_track_end_to_end_sweep(nets, syn_mesh, case):
    """Track against the oracle chain on frames other than the default scene: the synthetic scene family (object 0.55-0.95 m away, off axis,
    any orientation), a 1280x720 frame, a starting pose 12 deg / 3 cm off, two iterations"""
    W, H = (1280, 720) if case == 4 else (640, 480)
    scene = syn.heldout_scenes(syn_mesh, 6, W=W, H=H)[case]
    hyp = syn.perturb_pose(scene.gt_pose, deg=12.0 if case == 5 else 5.0, trans=0.03 if case == 5 else 0.01, seed=50 + case)
    itr = 2 if case == 3 else 1
    m = FoundationPose(syn_mesh, scene.K, nets[0], nets[1])
    try:
        ok, pose = m.Track(scene.rgb, scene.depth, hyp, syn_mesh.name, refine_itr=itr)
        assert ok, m.last_error
    finally:
        m.close()
    om = fo.OracleMesh(syn_mesh)
    p16 = syn.to_colmajor(hyp[None])
    for _ in range(itr):
        a = fo.render(om, p16, scene.K, (H, W), 1.2)
        b = fo.crop(scene.rgb, scene.depth, scene.K, p16, 1.2, syn_mesh.diameter)
        with torch.no_grad():
            t, r = nets[2](torch.from_numpy(a), torch.from_numpy(b))
        p16 = fo.refine_post_process(p16, t.numpy(), r.numpy(), syn_mesh.diameter)
    ang, dist = _pose_err(pose, syn.from_colmajor(p16)[0])
    assert ang < 0.1 * itr and dist < 1e-4 * itr, (ang, dist)      # (the second iteration renders from the first one's f16-rounded pose)


def test_register_end_to_end_252(model, nets, syn_mesh, syn_scene):
    ok, pose = model.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)
    assert ok, model.last_error
    refined, scores = _oracle_register(nets, syn_mesh, syn_scene, 252)
    # the returned pose must be one of the oracle's refined hypotheses (within fp16 noise) ...
    errs = [_pose_err(pose, r) for r in refined]
    idx = int(np.argmin([e[0] + 1e3 * e[1] for e in errs]))
    assert errs[idx][0] < 0.1 and errs[idx][1] < 1e-4, errs[idx]
    # ... and its oracle score must be the oracle's maximum up to the score tolerance (ties broken by fp16 noise)
    assert scores[idx] >= scores.max() - 5e-3, (idx, scores[idx], scores.max(), int(scores.argmax()))
    # mismatching sizes -> False like CheckInputArguments
    ok, _ = model.Register(syn_scene.rgb, syn_scene.depth[:100], syn_scene.mask, syn_mesh.name)
    assert not ok


@pytest.mark.parametrize("steps", [1, 2])
def test_register_mid_sized_batches_match_oracle(model, nets, syn_mesh, syn_scene, steps):
    """42 / 84 hypotheses (in-plane steps 1 / 2): the batch sizes where the schedule choice changes layer by layer
    (resident-halo kernels above ~32 hypotheses, implicit-GEMM tiles below, split-K off)."""
    n = 42 * steps
    model.set_inplane_steps(steps)
    try:
        assert model.num_hypotheses == n
        ok, pose = model.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)
    finally:
        model.set_inplane_steps(6)
    assert ok, model.last_error
    om = fo.OracleMesh(syn_mesh)
    poses = fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, syn_scene.K, inplane_step=360 // steps)
    assert len(poses) == n
    a = fo.render(om, poses, syn_scene.K, syn_scene.depth.shape, 1.2)
    b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, poses, 1.2, syn_mesh.diameter)
    with torch.no_grad():
        t, r = nets[2](torch.from_numpy(a), torch.from_numpy(b))
    refined = fo.refine_post_process(poses, t.numpy(), r.numpy(), syn_mesh.diameter)
    a = fo.render(om, refined, syn_scene.K, syn_scene.depth.shape, 1.1)
    b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, refined, 1.1, syn_mesh.diameter)
    with torch.no_grad():
        scores = nets[3](torch.from_numpy(a), torch.from_numpy(b)).numpy()
    errs = [_pose_err(pose, x) for x in syn.from_colmajor(refined)]
    idx = int(np.argmin([e[0] + 1e3 * e[1] for e in errs]))
    assert errs[idx][0] < 0.1 and errs[idx][1] < 1e-4, errs[idx]
    assert scores[idx] >= scores.max() - 5e-3, (idx, scores[idx], scores.max())


def test_register_two_refine_iterations_matches_oracle(model, nets, syn_mesh, syn_scene):
    """refine_itr = 2: iteration 0 uses the shared observed crop (one translation for all hypotheses), iteration 1 the
    per-hypothesis crops; both must agree with the oracle pipeline that always computes every crop."""
    n = 252
    ok, pose = model.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name, refine_itr=2)
    assert ok, model.last_error
    om = fo.OracleMesh(syn_mesh)
    poses = fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, syn_scene.K)[:n]
    for _ in range(2):
        a = fo.render(om, poses, syn_scene.K, syn_scene.depth.shape, 1.2)
        b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, poses, 1.2, syn_mesh.diameter)
        with torch.no_grad():
            t, r = nets[2](torch.from_numpy(a), torch.from_numpy(b))
        poses = fo.refine_post_process(poses, t.numpy(), r.numpy(), syn_mesh.diameter)
    a = fo.render(om, poses, syn_scene.K, syn_scene.depth.shape, 1.1)
    b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, poses, 1.1, syn_mesh.diameter)
    with torch.no_grad():
        s = nets[3](torch.from_numpy(a), torch.from_numpy(b)).numpy()
    refined = syn.from_colmajor(poses)
    errs = [_pose_err(pose, r) for r in refined]
    idx = int(np.argmin([e[0] + 1e3 * e[1] for e in errs]))
    assert errs[idx][0] < 0.2 and errs[idx][1] < 2e-4, errs[idx]
    assert s[idx] >= s.max() - 5e-3


def test_sharded_register_single_rank_matches_plain_register(model, syn_mesh, syn_scene):
    """the packed shard protocol through the torch.distributed helper (world size 1, RCCL backend) must return the same pose as
    fp_register; 2- and 8-shard emulations (one begin per rank into its slot of the gather buffer) must agree as well."""
    import socket
    import torch.distributed as dist
    from foundationpose_cpp_amd.distributed import HipShardBackend, shard_range, sharded_register
    ok, pose = model.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)
    assert ok
    dev = torch.device("cuda", 0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rgb, depth, mask = (torch.from_numpy(a).to(dev) for a in (syn_scene.rgb, syn_scene.depth, syn_scene.mask))
        be = HipShardBackend(model, dev)
        p16, idx = sharded_register(be, dist, 252, rgb, depth, mask, 480, 640, syn_mesh.name, 1)
        np.testing.assert_allclose(syn.from_colmajor(p16), pose, atol=1e-6)
        # emulate two and eight ranks on one GPU: every rank's packed rows computed one after the other into its slot of the
        # gather buffer (what the all-gather would deliver), then the redundant finish -- incl. the ragged last shard of 252 / 8
        rows = {}
        for world in (2, 8):
            per = -(-252 // world)
            packed, gathered = be.buffers(per, world)
            for r in range(world):
                b, c = shard_range(252, world, r)
                be.shard_begin_packed(rgb, depth, mask, 480, 640, syn_mesh.name, 1, b, c, packed, per)
                be.before_collective()
                gathered[r * per:(r + 1) * per].copy_(packed)
                be.after_collective()
            p16b, idxb = be.shard_finish_packed(gathered, 252)
            rows[world] = gathered[:252].cpu().numpy()
            # the returned pose is the gathered pose of the returned hypothesis
            np.testing.assert_array_equal(p16b, rows[world][idxb, 512:])
            if world == 2:      # slices of 126 run the same schedules as the full batch: same winner, same pose
                assert idxb == idx, (world, idxb, idx)
                np.testing.assert_allclose(p16b, p16, atol=1e-6)
        # slices of 32 take other schedules (implicit-GEMM tiles instead of the resident-halo kernels): the same values up to
        # fp32 summation order -- refined poses agree to 1e-4, pooled features to 2e-3 of their scale -- but with synthetic
        # weights the 252 scores are tied to ~1e-5, so the arg-max itself may differ between shardings
        np.testing.assert_allclose(rows[8][:, 512:], rows[2][:, 512:], atol=1e-4)
        fscale = np.abs(rows[2][:, :512]).max()
        assert np.abs(rows[8][:, :512] - rows[2][:, :512]).max() < 2e-3 * fscale
        # a failing sampler (empty mask) is reported by the finish of the rank that ran it, the caller's pose stays untouched
        packed, gathered = be.buffers(252, 1)
        be.shard_begin_packed(rgb, depth, torch.zeros_like(mask), 480, 640, syn_mesh.name, 1, 0, 252, packed, 252)
        be.before_collective(); gathered.copy_(packed); be.after_collective()
        with pytest.raises(Exception) as e:
            be.shard_finish_packed(gathered, 252)
        assert "Mask is all zero" in str(e.value)
    finally:
        dist.destroy_process_group()


def test_register_is_deterministic(model, syn_mesh, syn_scene):
    poses = [model.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)[1] for _ in range(3)]
    np.testing.assert_array_equal(poses[0], poses[1])
    np.testing.assert_array_equal(poses[0], poses[2])


def _track_both_ways(mesh, scene, nets, hyp, Wd, H):
    """Track from a HOST frame on a model that has never seen a whole frame (only the rows of the crop window are uploaded) vs Track
    reading the whole frame in place from device memory"""
    m1 = FoundationPose(mesh, scene.K, nets[0], nets[1], max_input_image_height=max(1080, H), max_input_image_width=max(1920, Wd))
    m2 = FoundationPose(mesh, scene.K, nets[0], nets[1], max_input_image_height=max(1080, H), max_input_image_width=max(1920, Wd))
    try:
        ok1, p1 = m1.Track(scene.rgb, scene.depth, hyp, mesh.name)
        assert ok1, m1.last_error
        rgb, depth = torch.from_numpy(scene.rgb).cuda(), torch.from_numpy(scene.depth).cuda()
        out = np.zeros(16, np.float32)
        h16 = syn.to_colmajor(hyp[None].astype(np.float32))[0]
        m2._must(m2._L.fp_track_ex(m2.handle, C.c_void_p(rgb.data_ptr()), C.c_void_p(depth.data_ptr()), 1, H, Wd, _p(h16),
                                   mesh.name.encode(), 1, _p(out)))
        return p1, syn.from_colmajor(out[None])[0]
    finally:
        m1.close()
        m2.close()


@pytest.mark.parametrize("Wd,H,ty", [(640, 480, -0.45), (640, 480, 0.45), (640, 480, -0.9), (640, 480, 0.9), (1280, 720, -0.02), (1280, 720, 0.385),
                                     (1280, 720, -0.39)])
def test_track_partial_row_upload_at_the_image_border(nets, syn_mesh, Wd, H, ty):
    """the host-side estimate of the observed-crop window (fp_api.hip track_submit_impl: double precision + margin) against what the
    kernels read: windows crossing row 0 / row H-1, windows entirely outside the frame (nothing is uploaded), 1280x720"""
    scene = syn.make_scene(syn_mesh, Wd, H)
    hyp = syn.perturb_pose(scene.gt_pose)
    hyp[1, 3] = ty
    v0 = scene.K[1, 2] + scene.K[1, 1] * ty / hyp[2, 3]
    rad = scene.K[1, 1] * syn_mesh.diameter * 0.6 / hyp[2, 3]
    kind = "outside" if (v0 + rad < 0 or v0 - rad > H) else ("border" if (v0 - rad < 0 or v0 + rad > H - 1) else "inside")
    assert kind == ("outside" if abs(ty) > 0.8 else "inside" if abs(ty) < 0.1 else "border"), (kind, v0, rad)
    p1, p2 = _track_both_ways(syn_mesh, scene, nets, hyp, Wd, H)
    np.testing.assert_array_equal(p1, p2)


def test_track_partial_row_upload_random_windows(nets, syn_mesh):
    """60 random hypotheses (depth 0.3-1.5 m, rows far beyond both borders) on FRESH noise frames, one reused model: a row the host
    estimate misses would still hold the previous frame's noise and change the pose"""
    rng = np.random.default_rng(12)
    K = syn.intrinsics()
    m1 = FoundationPose(syn_mesh, K, nets[0], nets[1])
    m2 = FoundationPose(syn_mesh, K, nets[0], nets[1])
    base = syn.perturb_pose(syn.pose_matrix(syn.random_rotation(3), [0, 0, 0.7]).astype(np.float32))
    out = np.zeros(16, np.float32)
    try:
        for k in range(60):
            rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
            depth = rng.uniform(0.2, 2.0, (480, 640)).astype(np.float32)
            hyp = base.copy()
            tz = rng.uniform(0.3, 1.5)
            hyp[:3, 3] = [rng.uniform(-0.5, 0.5) * tz, rng.uniform(-1.2, 1.2) * tz, tz]
            ok, p1 = m1.Track(rgb, depth, hyp, syn_mesh.name)
            assert ok, m1.last_error
            r_d, d_d = torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda()
            h16 = syn.to_colmajor(hyp[None])[0]
            m2._must(m2._L.fp_track_ex(m2.handle, C.c_void_p(r_d.data_ptr()), C.c_void_p(d_d.data_ptr()), 1, 480, 640, _p(h16),
                                       syn_mesh.name.encode(), 1, _p(out)))
            np.testing.assert_array_equal(p1, syn.from_colmajor(out[None])[0], err_msg=f"iteration {k}, t = {hyp[:3, 3]}")
        # degenerate hypotheses: behind the camera / absurd translations must not crash the host-side window arithmetic
        for t in ([0, 0, -0.5], [0, 1e30, 1e-5], [0, -1e30, 0.5], [0, 0, 1e-12]):
            hyp = base.copy()
            hyp[:3, 3] = t
            ok, p1 = m1.Track(rgb, depth, hyp, syn_mesh.name)
            m2._must(m2._L.fp_track_ex(m2.handle, C.c_void_p(r_d.data_ptr()), C.c_void_p(d_d.data_ptr()), 1, 480, 640,
                                       _p(syn.to_colmajor(hyp[None])[0]), syn_mesh.name.encode(), 1, _p(out)))
            assert ok
            a, b = p1, syn.from_colmajor(out[None])[0]
            assert np.array_equal(a, b) or (np.isnan(a) == np.isnan(b)).all(), t
    finally:
        m1.close()
        m2.close()


def test_track_window_upload_alternates_with_whole_frames(nets, syn_mesh):
    """[r4] ONE model served in turn from host frames (the packed crop window + its own frame record, one copy), from frames resident
    in device memory, from host frames whose window is too wide to be packed (whole rows), and by a Register in between: every Track
    equals the Track of a second model that only ever reads whole device frames -- the frame record the replayed graph reads is
    re-published whenever the source changes"""
    rng = np.random.default_rng(5)
    K = syn.intrinsics()
    m1 = FoundationPose(syn_mesh, K, nets[0], nets[1])
    m2 = FoundationPose(syn_mesh, K, nets[0], nets[1])
    scene = syn.make_scene(syn_mesh)
    base = syn.perturb_pose(syn.pose_matrix(syn.random_rotation(3), [0, 0, 0.7]).astype(np.float32))
    out = np.zeros(16, np.float32)

    def device_track(m, rgb, depth, hyp):
        r_d, d_d = torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda()
        m._must(m._L.fp_track_ex(m.handle, C.c_void_p(r_d.data_ptr()), C.c_void_p(d_d.data_ptr()), 1, 480, 640,
                                 _p(syn.to_colmajor(hyp[None])[0]), syn_mesh.name.encode(), 1, _p(out)))
        return syn.from_colmajor(out[None])[0].copy()

    try:
        for k, kind in enumerate(["host", "host", "device", "host", "wide", "host", "register", "host", "device", "wide", "host", "host"]):
            rgb = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
            depth = rng.uniform(0.2, 2.0, (480, 640)).astype(np.float32)
            hyp = base.copy()
            tz = 0.25 if kind == "wide" else rng.uniform(0.5, 1.2)      # 0.25 m: the crop window is wider than half the frame
            hyp[:3, 3] = [rng.uniform(-0.2, 0.2) * tz, rng.uniform(-0.2, 0.2) * tz, tz]
            if kind == "register":
                ok, _ = m1.Register(scene.rgb, scene.depth, scene.mask, syn_mesh.name)
                assert ok, m1.last_error
                continue
            if kind == "device":
                p1 = device_track(m1, rgb, depth, hyp)
            else:
                ok, p1 = m1.Track(rgb, depth, hyp, syn_mesh.name)
                assert ok, m1.last_error
            np.testing.assert_array_equal(p1, device_track(m2, rgb, depth, hyp), err_msg=f"step {k} ({kind})")
    finally:
        m1.close()
        m2.close()


def test_native_sharded_register_world_1(model, syn_mesh, syn_scene):
    """fp_register_sharded through the Python host: an RCCL communicator of size 1 made with the process's librccl (NativeRcclComm), the
    library's own begin -> exchange -> finish on its stream; the result is the unsharded Register's, bit for bit.  (The C++ twin with
    ncclCommInitAll is examples/fp_demo_mgpu.cpp, tests/test_demo_gpu.py; the arithmetic of N > 1 is covered by the gloo tests.)"""
    import torch.distributed as dist
    from foundationpose_cpp_amd.distributed import NativeRcclComm, sharded_register_native
    dev = torch.device("cuda", 0)
    comm = NativeRcclComm(dist, dev)
    try:
        assert comm.world == 1 and comm.rank == 0
        rgb, depth, mask = (torch.from_numpy(x).to(dev) for x in (syn_scene.rgb, syn_scene.depth, syn_scene.mask))
        ok, ref = model.Register(syn_scene.rgb, syn_scene.depth, syn_scene.mask, syn_mesh.name)
        assert ok, model.last_error
        for _ in range(3):      # eager, capture, replay
            pose16, idx = sharded_register_native(model, comm, rgb, depth, mask, 480, 640, syn_mesh.name)
            np.testing.assert_array_equal(syn.from_colmajor(pose16[None])[0], ref)
            assert 0 <= idx < model.num_hypotheses
        # a bad mask fails like the unsharded call, and the model stays usable
        with pytest.raises(Exception) as e:
            sharded_register_native(model, comm, rgb, depth, torch.zeros_like(mask), 480, 640, syn_mesh.name)
        assert "Mask is all zero" in str(e.value)
        pose16, _ = sharded_register_native(model, comm, rgb, depth, mask, 480, 640, syn_mesh.name)
        np.testing.assert_array_equal(syn.from_colmajor(pose16[None])[0], ref)
    finally:
        comm.close()
