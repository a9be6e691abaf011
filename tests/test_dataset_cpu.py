"""Frame / dataset I/O (C ABI fp_read_rgb_depth_mask, fp_read_cam_k, fp_image_*; reference helpers
simple_tests/include/tests/help_func.hpp:10-129) against PIL / numpy.  Host code only: runs on CPU."""
import ctypes as C
import os

import numpy as np
import pytest
from PIL import Image

from foundationpose_cpp_amd import _lib, dataset as D, load_mesh, synthetic as syn


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_synthetic_sequence_round_trip(tmp_path):
    root = str(tmp_path / "seq0")
    mesh = syn.make_mesh(subdiv=2)
    mesh, gts = D.write_synthetic_sequence(root, n_frames=3, mesh=mesh)
    seq = D.Sequence(root)
    assert len(seq) == 3 and (seq.H, seq.W) == (480, 640)
    np.testing.assert_allclose(seq.K, syn.intrinsics(), rtol=1e-7)
    for i in range(3):
        sc = syn.make_scene(mesh, pose=gts[i])
        rgb, depth, mask = seq.frame(i, with_mask=True)
        np.testing.assert_array_equal(rgb, sc.rgb)
        # u16 millimetres: what the reference's depth PNGs hold (help_func.hpp:22-23)
        np.testing.assert_array_equal(depth, np.clip(np.rint(sc.depth * 1000.0), 0, 65535).astype(np.uint16).astype(np.float32) / 1000.0)
        np.testing.assert_array_equal(mask > 0, sc.mask > 0)
        # independent decoder agrees
        np.testing.assert_array_equal(rgb, np.asarray(Image.open(seq._path("rgb", i)).convert("RGB")))
        np.testing.assert_array_equal((depth * 1000.0 + 0.5).astype(np.uint16), np.asarray(Image.open(seq._path("depth", i))).astype(np.uint16))
    m2 = load_mesh("obj", seq.mesh_path())
    assert m2.vertices.shape == mesh.vertices.shape and abs(m2.diameter - mesh.diameter) < 1e-6


def test_mask_channel_rule_and_png_variants(tmp_path):
    """3-channel masks keep the first channel after BGR2RGB = the file's R channel (help_func.hpp:26-32);
    16-bit RGB, palette and filter types decode like PIL."""
    L = _lib.lib()
    rng = np.random.default_rng(0)
    H, W = 37, 53
    rgbm = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    Image.fromarray(rgbm).save(tmp_path / "m.png")
    Image.fromarray(rgbm).save(tmp_path / "r.png")
    d16 = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    Image.fromarray(d16).save(tmp_path / "d.png")
    rgb, depth, mask = np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.float32), np.zeros((H, W), np.uint8)
    assert L.fp_read_rgb_depth_mask(str(tmp_path / "r.png").encode(), str(tmp_path / "d.png").encode(),
                                    str(tmp_path / "m.png").encode(), H, W, _p(rgb), _p(depth), _p(mask)) == 0, _lib.last_error()
    np.testing.assert_array_equal(rgb, rgbm)
    np.testing.assert_array_equal(mask, rgbm[..., 0])
    np.testing.assert_array_equal(depth, d16.astype(np.float32) / 1000.0)
    # smooth gradients make PIL's encoder pick Sub / Up / Average / Paeth filters
    yy, xx = np.mgrid[0:H, 0:W]
    grad = np.stack([(xx * 4) % 256, (yy * 6) % 256, ((xx + yy) * 3) % 256], -1).astype(np.uint8)
    for mode, name in (("RGB", "g.png"), ("RGBA", "ga.png"), ("L", "gl.png"), ("P", "gp.png"), ("LA", "gla.png")):
        Image.fromarray(grad).convert(mode).save(tmp_path / name, optimize=True)
        h, w, ch, bits = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert L.fp_image_read_png(str(tmp_path / name).encode(), C.byref(h), C.byref(w), C.byref(ch), C.byref(bits), None, 0) == 0
        out = np.zeros((h.value, w.value, ch.value), np.uint16)
        assert L.fp_image_read_png(str(tmp_path / name).encode(), None, None, None, None, _p(out), out.size) == 0
        ref = np.asarray(Image.open(tmp_path / name).convert("RGB" if mode == "P" else mode))
        np.testing.assert_array_equal(out.reshape(ref.shape), ref)
    # rgb frames in the other containers cv::imread reads (JPEG, BMP): identical to PIL's decode; fp_frame_size too
    for name, kw in (("r.jpg", {"quality": 90}), ("r.bmp", {})):
        Image.fromarray(rgbm).save(tmp_path / name, **kw)
        assert L.fp_read_rgb_depth_mask(str(tmp_path / name).encode(), None, None, H, W, _p(rgb), None, None) == 0, _lib.last_error()
        np.testing.assert_array_equal(rgb, np.asarray(Image.open(tmp_path / name).convert("RGB")))
        h, w = C.c_int(), C.c_int()
        assert L.fp_frame_size(str(tmp_path / name).encode(), C.byref(h), C.byref(w)) == 0 and (h.value, w.value) == (H, W)
    # errors: reference CHECKs with "Failed reading ... from path" / "Failed open file"
    assert L.fp_read_rgb_depth_mask(b"/nonexistent.png", None, None, H, W, _p(rgb), None, None) != 0
    assert "Failed reading rgb from path" in _lib.last_error()
    K = np.zeros(9, np.float32)
    assert L.fp_read_cam_k(b"/nonexistent.txt", _p(K)) != 0 and "Failed open file" in _lib.last_error()


def test_write_png_and_bbox_overlay(tmp_path):
    L = _lib.lib()
    img = np.zeros((120, 160, 3), np.uint8)
    K = np.array([[100, 0, 80], [0, 100, 60], [0, 0, 1]], np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = 1.0
    out = D.draw_bbox3d(img, K, pose, [0.4, 0.2, 0.2])
    green = (out == np.array([0, 255, 0], np.uint8)).all(-1)
    assert green.sum() > 300
    # front face (z = 1 - 0.1): corners at u = 80 +- 100*0.2/0.9, v = 60 +- 100*0.1/0.9
    for u, v in ((80 - 22.2, 60 - 11.1), (80 + 22.2, 60 + 11.1)):
        assert green[int(round(v)) - 1:int(round(v)) + 3, int(round(u)) - 1:int(round(u)) + 3].any()
    assert not green[60, 80]                                  # box interior untouched
    p = str(tmp_path / "o.png")
    assert L.fp_image_write_png_rgb(p.encode(), _p(out), 120, 160) == 0
    np.testing.assert_array_equal(np.asarray(Image.open(p)), out)
