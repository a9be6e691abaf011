"""The CPU oracle against the committed known-answer vectors (tests/golden/fp_golden_v1.npz, made by
tests/golden/make_golden.py).  Geometry is bit-exact (same C code, -ffp-contract=off, same libm); the PyTorch fp32
networks are compared with a tolerance because oneDNN's reduction order depends on the host's thread count."""
import hashlib
import os

import numpy as np
import pytest
import torch

from foundationpose_cpp_amd import synthetic as syn, weights as W
from oracle import fp_oracle as fo
from oracle import nets_torch as NT

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp_golden_v1.npz")
NN_TOL = dict(rtol=0, atol=2e-5)     # trans / rot outputs are O(0.1); fp32 summation-order noise only


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_synthetic_inputs_are_the_ones_the_vectors_were_made_for(gold, syn_mesh, syn_scene):
    assert _sha(syn_scene.rgb) == str(gold["sha_rgb"])
    assert _sha(syn_scene.depth) == str(gold["sha_depth"])
    assert _sha(syn_scene.mask) == str(gold["sha_mask"])
    assert _sha(syn_mesh.vertices) == str(gold["sha_vertices"])
    assert _sha(syn_mesh.texture) == str(gold["sha_texture"])
    np.testing.assert_array_equal(syn_scene.K.astype(np.float32), gold["K"])
    assert np.float32(syn_mesh.diameter) == gold["diameter"]


def test_sampler_vectors(gold, syn_scene):
    np.testing.assert_array_equal(fo.rotation_grid(40, 60), gold["rotation_grid_252"])
    g24 = fo.rotation_grid(40, 15)
    assert len(g24) == 1008
    np.testing.assert_array_equal(g24[[0, 1, 23, 24, 1007]], gold["rotation_grid_1008_first_last"])
    poses = fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, syn_scene.K)
    np.testing.assert_array_equal(poses[0, 12:15], gold["hyp_center"])
    er = fo.erode_depth(syn_scene.depth)
    np.testing.assert_array_equal(er[200:280:8], gold["depth_eroded_rows"])
    np.testing.assert_array_equal(fo.bilateral_filter_depth(er)[200:280:8], gold["depth_bilateral_rows"])
    np.testing.assert_array_equal(fo.depth_to_xyz(syn_scene.depth, syn_scene.K)[200:280:8], gold["xyz_rows"])


@pytest.mark.parametrize("ratio,tag", [(1.2, "r12"), (1.1, "r11")])
def test_render_and_crop_vectors(gold, syn_mesh, syn_scene, ratio, tag):
    poses = fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, syn_scene.K)[gold["hyp_ids"]]
    np.testing.assert_array_equal(fo.crop_window_tf(poses, syn_scene.K, ratio, syn_mesh.diameter), gold[f"crop_tf_{tag}"])
    a, tri, _ = fo.render(fo.OracleMesh(syn_mesh), poses, syn_scene.K, syn_scene.depth.shape, ratio, debug=True)
    np.testing.assert_array_equal(tri, gold[f"tri_id_{tag}"])
    np.testing.assert_array_equal(a, gold[f"render_{tag}"])
    b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, poses, ratio, syn_mesh.diameter)
    np.testing.assert_array_equal(b, gold[f"transf_{tag}"])


def test_network_and_pipeline_vectors(gold, syn_mesh, syn_scene, tmp_path):
    rs = W.pack_synthetic("refiner", str(tmp_path / "r.fpw"))
    ss = W.pack_synthetic("scorer", str(tmp_path / "s.fpw"))
    refiner, scorer = NT.build("refiner", rs).eval(), NT.build("scorer", ss).eval()
    om, K, hw, diam = fo.OracleMesh(syn_mesh), syn_scene.K, syn_scene.depth.shape, syn_mesh.diameter
    poses = fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, K)
    with torch.no_grad():
        sel = poses[gold["nn_ids"]]
        t, r = refiner(torch.from_numpy(fo.render(om, sel, K, hw, 1.2)),
                       torch.from_numpy(fo.crop(syn_scene.rgb, syn_scene.depth, K, sel, 1.2, diam)))
        np.testing.assert_allclose(t.numpy(), gold["refiner_trans"], **NN_TOL)
        np.testing.assert_allclose(r.numpy(), gold["refiner_rot"], **NN_TOL)
        np.testing.assert_allclose(fo.refine_post_process(sel, gold["refiner_trans"], gold["refiner_rot"], diam),
                                   gold["refined_poses"], rtol=0, atol=1e-7)
        # configs[0]: Register over the first 8 hypotheses; the scorer runs on the STORED refined poses so that fp32
        # noise in the refiner cannot move a crop window by a pixel
        p8r = gold["register8_refined"]
        sc = scorer(torch.from_numpy(fo.render(om, p8r, K, hw, 1.1)),
                    torch.from_numpy(fo.crop(syn_scene.rgb, syn_scene.depth, K, p8r, 1.1, diam))).numpy().reshape(-1)
        np.testing.assert_allclose(sc, gold["register8_scores"], rtol=0, atol=1e-4)
        assert fo.argmax(gold["register8_scores"]) == int(gold["register8_best"])
        hyp = gold["track_in"]
        t, r = refiner(torch.from_numpy(fo.render(om, hyp, K, hw, 1.2)),
                       torch.from_numpy(fo.crop(syn_scene.rgb, syn_scene.depth, K, hyp, 1.2, diam)))
        np.testing.assert_allclose(fo.refine_post_process(hyp, t.numpy(), r.numpy(), diam), gold["track_out"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(syn.to_colmajor(syn.perturb_pose(syn_scene.gt_pose)[None]), hyp)


def test_discriminating_weight_set_vectors(disc_nets, syn_mesh, syn_scene):
    """tests/golden/fp_golden_disc_v1.npz: 42 hypotheses under the discriminating weights (make_golden.py wrote it)"""
    g = np.load(os.path.join(os.path.dirname(GOLD), "fp_golden_disc_v1.npz"))
    om, K, hw, diam = fo.OracleMesh(syn_mesh), syn_scene.K, syn_scene.depth.shape, syn_mesh.diameter
    sel = fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, K)[::int(g["hyp_step"])]
    with torch.no_grad():
        t, r = disc_nets[2](torch.from_numpy(fo.render(om, sel, K, hw, 1.2)),
                            torch.from_numpy(fo.crop(syn_scene.rgb, syn_scene.depth, K, sel, 1.2, diam)))
        # fp32 torch on another host / thread count: 1e-3 of the between-hypothesis spread
        for got, ref in ((t.numpy(), g["refiner_trans"]), (r.numpy(), g["refiner_rot"])):
            assert np.abs(got - ref).max() <= 2e-3 * ref.std(0).min(), np.abs(got - ref).max() / ref.std(0).min()
        ref = g["refined_poses"]
        sc = disc_nets[3](torch.from_numpy(fo.render(om, ref, K, hw, 1.1)),
                          torch.from_numpy(fo.crop(syn_scene.rgb, syn_scene.depth, K, ref, 1.1, diam))).numpy()
    assert np.abs(sc - g["scores"]).max() <= 2e-3 * g["scores"].std()
    assert g["scores"].std() >= 0.5 and fo.argmax(g["scores"]) == int(g["best"])
