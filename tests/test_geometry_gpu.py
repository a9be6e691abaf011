"""Parity of the HIP geometry path (C ABI) against the CPU oracle on the same seeded inputs.

Bar: integer work (the rasteriser's triangle-id buffer) is BIT-EXACT; float tensors are compared with the tolerance
written at each assert (both sides evaluate the same float expression order; residual differences come from
expf/tanhf/sinf implementations and are <= a few ulp).
"""
import numpy as np
import pytest

from foundationpose_cpp_amd import FoundationPose, synthetic as syn
from oracle import fp_oracle as fo

pytestmark = pytest.mark.gpu

F32_TOL = dict(rtol=0, atol=2e-6)   # render / crop tensors live in [-4,4]; 2e-6 abs ~ a few ulp


@pytest.fixture(scope="module")
def model(syn_mesh):
    m = FoundationPose(syn_mesh, syn.intrinsics())
    yield m
    m.close()


@pytest.fixture(scope="module")
def hyp(model, syn_scene):
    model.upload_frame(syn_scene.rgb, syn_scene.depth)
    poses = model.get_hyp_poses(syn_scene.mask)
    assert poses is not None and poses.shape == (252, 4, 4)
    return poses


def test_rotation_grid_and_translation_match_oracle(hyp, syn_scene):
    ref = syn.from_colmajor(fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, syn_scene.K))
    np.testing.assert_allclose(hyp[:, :3, :3], ref[:, :3, :3], atol=1e-6)
    # GuessTranslation runs on the device (exact radix-select median, the host code's float expression order): bit-exact
    np.testing.assert_array_equal(hyp[:, :3, 3], ref[:, :3, 3])


def test_depth_filters_and_xyz(model, syn_scene):
    model.upload_frame(syn_scene.rgb, syn_scene.depth)
    e, b = model.filter_depth()
    np.testing.assert_array_equal(e, fo.erode_depth(syn_scene.depth))      # compare / count only: bit exact
    np.testing.assert_allclose(b, fo.bilateral_filter_depth(fo.erode_depth(syn_scene.depth)), rtol=2e-6, atol=0)
    np.testing.assert_array_equal(model.xyz_map(), fo.depth_to_xyz(syn_scene.depth, syn_scene.K))


@pytest.mark.parametrize("crop_ratio", [1.2, 1.1])
def test_render_and_transform_parity_252(model, hyp, syn_mesh, syn_scene, crop_ratio):
    model.upload_frame(syn_scene.rgb, syn_scene.depth)
    om = fo.OracleMesh(syn_mesh)
    p16 = syn.to_colmajor(hyp)
    ref_a, ref_tri, ref_rast = fo.render(om, p16, syn_scene.K, (480, 640), crop_ratio, debug=True)
    ref_b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, p16, crop_ratio, syn_mesh.diameter)
    tri, rast = model.debug_rasterize(syn_mesh.name, hyp, crop_ratio)
    assert (ref_tri > 0).sum() > 252 * 2000
    np.testing.assert_array_equal(tri, ref_tri)                             # integer raster: bit exact
    np.testing.assert_allclose(rast, ref_rast, rtol=0, atol=2e-6)
    a, b = model.render_and_transform(syn_mesh.name, hyp, crop_ratio)
    np.testing.assert_allclose(a, ref_a, **F32_TOL)
    np.testing.assert_allclose(b, ref_b, **F32_TOL)


def test_render_gt_pose_and_untextured(syn_scene):
    mesh = syn.make_mesh(textured=False, name="plain")
    m = FoundationPose(mesh, syn.intrinsics())
    m.upload_frame(syn_scene.rgb, syn_scene.depth)
    poses = np.stack([syn_scene.gt_pose, syn.perturb_pose(syn_scene.gt_pose)])
    a, b = m.render_and_transform("plain", poses, 1.2)
    ref_a = fo.render(fo.OracleMesh(mesh), syn.to_colmajor(poses), syn_scene.K, (480, 640), 1.2)
    np.testing.assert_allclose(a, ref_a, **F32_TOL)
    m.close()


def test_render_edge_cases(syn_mesh, syn_scene):
    """object partly outside the image / crossing the near plane (frustum-clip path) / far away / 1 hypothesis"""
    m = FoundationPose(syn_mesh, syn.intrinsics())
    m.upload_frame(syn_scene.rgb, syn_scene.depth)
    om = fo.OracleMesh(syn_mesh)
    R = syn.random_rotation(11)
    cases = [(0.45, 0.3, 0.7), (0.0, 0.0, 0.12), (0.0, 0.0, 0.05), (0.3, -0.2, 5.0), (-0.6, 0.0, 0.7)]
    poses = np.stack([syn.pose_matrix(R, t) for t in cases])
    for i in range(len(poses)):                      # also exercises N=1
        p = poses[i:i + 1]
        tri, _ = m.debug_rasterize(syn_mesh.name, p, 1.2)
        a, b = m.render_and_transform(syn_mesh.name, p, 1.2)
        ra, rtri, _ = fo.render(om, syn.to_colmajor(p), syn_scene.K, (480, 640), 1.2, debug=True)
        rb = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, syn.to_colmajor(p), 1.2, syn_mesh.diameter)
        np.testing.assert_array_equal(tri, rtri)
        np.testing.assert_allclose(a, ra, **F32_TOL)
        np.testing.assert_allclose(b, rb, **F32_TOL)
    m.close()


def test_big_triangle_clip_path_bit_exact(syn_scene):
    v = np.array([[-5, -5, 2.0], [5, -5, 2.0], [0, 5, -3.0], [0.3, 0.3, 0.1]], np.float32)
    mesh = syn.Mesh("big", v, np.tile(np.array([[0, 0, -1]], np.float32), (4, 1)), np.zeros((4, 2), np.float32),
                    np.array([[0, 1, 2], [0, 1, 3]], np.int32), np.full((2, 2, 3), 100, np.uint8), diameter=0.2,
                    center=np.zeros(3, np.float32))
    m = FoundationPose(mesh, syn.intrinsics())
    m.upload_frame(syn_scene.rgb, syn_scene.depth)
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = 1.0
    tri, _ = m.debug_rasterize("big", pose[None], 1.2)
    _, rtri, _ = fo.render(fo.OracleMesh(mesh), syn.to_colmajor(pose[None]), syn_scene.K, (480, 640), 1.2, debug=True)
    assert (rtri > 0).sum() > 1000
    np.testing.assert_array_equal(tri, rtri)
    m.close()


def test_image_sizes_and_errors(syn_mesh):
    m = FoundationPose(syn_mesh, syn.intrinsics(1280, 720), max_input_image_height=720, max_input_image_width=1280)
    sc = syn.make_scene(syn_mesh, 1280, 720)
    m.upload_frame(sc.rgb, sc.depth)
    poses = m.get_hyp_poses(sc.mask)
    ref = syn.from_colmajor(fo.get_hyp_poses(sc.depth, sc.mask, sc.K))
    np.testing.assert_allclose(poses[:, :3, 3], ref[:, :3, 3], rtol=1e-5)
    a, b = m.render_and_transform(syn_mesh.name, poses[:8], 1.1)
    om = fo.OracleMesh(syn_mesh)
    np.testing.assert_allclose(a, fo.render(om, syn.to_colmajor(poses[:8]), sc.K, (720, 1280), 1.1), **F32_TOL)
    np.testing.assert_allclose(b, fo.crop(sc.rgb, sc.depth, sc.K, syn.to_colmajor(poses[:8]), 1.1, syn_mesh.diameter), **F32_TOL)
    # reference error behaviour: empty mask -> False (foundationpose_sampling.cpp:269), message kept
    assert m.get_hyp_poses(np.zeros_like(sc.mask)) is None and "Mask is all zero" in m.last_error
    # no valid depth under the mask (:278)
    m.upload_frame(sc.rgb, np.zeros_like(sc.depth))
    assert m.get_hyp_poses(sc.mask) is None and "No valid value" in m.last_error
    # oversize frame (CheckInputArguments foundationpose.cpp:171-172)
    big = np.zeros((1000, 1280), np.float32)
    with pytest.raises(Exception, match="unexpected size"):
        m.upload_frame(np.zeros((1000, 1280, 3), np.uint8), big)
    m.close()


def test_1008_hypothesis_grid(syn_mesh, syn_scene):
    m = FoundationPose(syn_mesh, syn.intrinsics())
    m.set_inplane_steps(24)
    assert m.num_hypotheses == 1008
    m.upload_frame(syn_scene.rgb, syn_scene.depth)
    poses = m.get_hyp_poses(syn_scene.mask)
    ref = syn.from_colmajor(fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, syn_scene.K, inplane_step=15))
    assert poses.shape == (1008, 4, 4)
    np.testing.assert_allclose(poses[:, :3, :3], ref[:, :3, :3], atol=1e-6)
    m.close()


def test_pose_update_and_argmax(model, syn_mesh):
    rng = np.random.default_rng(0)
    poses = np.stack([syn.pose_matrix(syn.random_rotation(i), rng.normal(size=3)) for i in range(300)])
    trans = rng.normal(size=(300, 3)).astype(np.float32)
    rot = (rng.normal(size=(300, 3)) * 2).astype(np.float32)
    rot[0] = 0
    trans[0] = 0
    out = model.refine_post_process(syn_mesh.name, poses, trans, rot)
    ref = syn.from_colmajor(fo.refine_post_process(syn.to_colmajor(poses), trans, rot, syn_mesh.diameter))
    np.testing.assert_allclose(out, ref, atol=1e-6)
    np.testing.assert_array_equal(out[0], poses[0])
    s = rng.normal(size=1008).astype(np.float32)
    s[[17, 500, 900]] = s.max() + 1
    assert model.argmax(s) == 17 == fo.argmax(s)
    assert model.argmax(s[:1]) == 0


def test_device_sampler_median_cases(syn_mesh):
    """GuessTranslation on the device against the oracle for odd / even counts, duplicates, one pixel and a full-frame
    mask (exact median: even counts average the two middle values; foundationpose_sampling.cpp:276-294)."""
    m = FoundationPose(syn_mesh, syn.intrinsics())
    rng = np.random.default_rng(12)
    H, W = 480, 640
    K = syn.intrinsics()
    rgb = np.zeros((H, W, 3), np.uint8)
    for case in range(6):
        depth = np.full((H, W), 0.8, np.float32)
        mask = np.zeros((H, W), np.uint8)
        if case == 0:      # odd count, distinct values
            mask[100:103, 200:211] = 1                     # 33 pixels
            depth[100:103, 200:211] = rng.uniform(0.5, 0.9, (3, 11)).astype(np.float32)
        elif case == 1:    # even count
            mask[50:54, 300:310] = 255                     # 40 pixels
            depth[50:54, 300:310] = rng.uniform(0.5, 0.9, (4, 10)).astype(np.float32)
        elif case == 2:    # many duplicates (quantised depth)
            mask[200:260, 100:180] = 1
            depth[200:260, 100:180] = (rng.integers(600, 610, (60, 80)) / 1000.0).astype(np.float32)
        elif case == 3:    # single pixel
            mask[7, 9] = 1
            depth[7, 9] = 0.731
        elif case == 4:    # full-frame mask with holes in the depth
            mask[:] = 1
            depth = rng.uniform(0.3, 1.2, (H, W)).astype(np.float32)
            depth[rng.uniform(size=(H, W)) < 0.3] = 0.0
        else:              # two blobs: bounding box spans both
            mask[10:20, 10:20] = 1
            mask[400:410, 600:620] = 1
            depth[:] = (0.5 + 0.0001 * np.arange(W, dtype=np.float32))[None, :]     # smooth ramp (noise would be eroded away)
        m.upload_frame(rgb, depth)
        got = m.get_hyp_poses(mask)
        ref = fo.get_hyp_poses(depth, mask, K)
        assert (got is None) == (ref is None), (case, m.last_error)
        assert got is not None, (case, m.last_error)
        np.testing.assert_array_equal(got[0, :3, 3], syn.from_colmajor(ref[:1])[0, :3, 3], err_msg=f"case {case}")
    m.close()


def test_separately_rounded_float_model_parity(syn_mesh, syn_scene):
    """The default float model of the rendering stage contracts multiply-adds like the reference's nvcc -fmad=true build (all
    tests above); FP_FLOAT_SEPARATE keeps every operation separately rounded.  HIP and oracle implement both: triangle ids
    bit-exact and tensors within F32_TOL in the second model too (252 hypotheses, edge / clip poses, the big-triangle clip case)."""
    m = FoundationPose(syn_mesh, syn.intrinsics())
    try:
        assert m.float_model == 1
        m.set_float_model(0)
        fo.set_fmad(False)
        m.upload_frame(syn_scene.rgb, syn_scene.depth)
        om = fo.OracleMesh(syn_mesh)
        poses = m.get_hyp_poses(syn_scene.mask)
        R = syn.random_rotation(11)
        edge = np.stack([syn.pose_matrix(R, t) for t in [(0.45, 0.3, 0.7), (0.0, 0.0, 0.12), (0.0, 0.0, 0.05), (-0.6, 0.0, 0.7)]])
        allp = np.concatenate([poses, edge])
        tri, _ = m.debug_rasterize(syn_mesh.name, allp, 1.2)
        a, _ = m.render_and_transform(syn_mesh.name, allp, 1.2)
        ra, rtri, _ = fo.render(om, syn.to_colmajor(allp), syn_scene.K, (480, 640), 1.2, debug=True)
        np.testing.assert_array_equal(tri, rtri)
        np.testing.assert_allclose(a, ra, **F32_TOL)
        # and the models differ from each other on the device exactly where they differ in the oracle
        m.set_float_model(1)
        fo.set_fmad(True)
        a1, _ = m.render_and_transform(syn_mesh.name, allp[:8], 1.2)
        r1 = fo.render(om, syn.to_colmajor(allp[:8]), syn_scene.K, (480, 640), 1.2)
        np.testing.assert_allclose(a1, r1, **F32_TOL)
        assert np.abs(a1 - a[:8]).max() > 0
    finally:
        fo.set_fmad(True)
        m.close()


# ---- seeded randomised sweep: meshes, intrinsics, frame sizes and poses nobody hand-picked --------------------------------------------
_SWEEP_SIZES = [(640, 480), (641, 479), (333, 257), (1280, 720), (1920, 1080), (160, 120)]


def _random_case(seed):
    """one frame + one mesh + 6 poses: a dented icosphere (1-3 subdivisions, own axes, NOT centred at the origin, texture coordinates
    beyond [0, 1] -> wrap addressing, odd texture size), fx != fy with the principal point off centre, an rgb frame of noise over a wavy depth surface with 10 % of
    it missing, objects from 0.12 m (near-plane clipping) to 3 m and up to a diameter outside the image"""
    rng = np.random.default_rng(1000 + seed)
    W, H = _SWEEP_SIZES[seed % len(_SWEEP_SIZES)]
    s, faces = syn._icosphere(1 + seed % 3)
    ax = rng.uniform(0.02, 0.12, 3)
    v = s * ax * (1.0 + 0.25 * rng.uniform(-1, 1, (len(s), 1))) + rng.uniform(-0.05, 0.05, 3)
    n = s / ax
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    uv = rng.uniform(-1.5, 2.5, (len(s), 2))
    tex = rng.integers(0, 256, (int(rng.integers(2, 70)), int(rng.integers(2, 70)), 3), dtype=np.uint8)
    mesh = syn.Mesh(f"rnd{seed}", v.astype(np.float32), n.astype(np.float32), uv.astype(np.float32), faces, tex).finalize()
    f = rng.uniform(0.6, 1.4) * W
    K = np.array([[f, 0, W * rng.uniform(0.3, 0.7)], [0, f * rng.uniform(0.8, 1.25), H * rng.uniform(0.3, 0.7)], [0, 0, 1]], np.float32)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (rng.uniform(0.3, 2.0) + 0.5 * np.sin(xx / W * rng.uniform(2, 9) + rng.uniform(0, 6)) * np.cos(yy / H * rng.uniform(2, 9)) * rng.uniform(0.1, 0.5)
             + rng.normal(0, 0.002, (H, W))).astype(np.float32)      # a smooth surface (the sampler's erosion removes isolated pixels) + 2 mm of noise
    depth[rng.uniform(size=(H, W)) < 0.1] = 0.0
    depth[:H // 8, :W // 8] = rng.uniform(0.0005, 6.0, (H // 8, W // 8))   # one corner of pure noise incl. values under min_depth and over the 4 m cut
    poses = []
    for i in range(6):
        z = float(rng.choice([0.12, 0.3, 0.7, 1.5, 3.0])) * rng.uniform(0.9, 1.1)
        u, w = rng.uniform(-0.1, 1.1) * W, rng.uniform(-0.1, 1.1) * H      # the centre projects up to 10 % outside the frame
        t = (z * (u - K[0, 2]) / K[0, 0], z * (w - K[1, 2]) / K[1, 1], z)
        poses.append(syn.pose_matrix(syn.random_rotation(100 * seed + i), t))
    return mesh, K, rgb, depth, np.stack(poses).astype(np.float32), (H, W)


@pytest.mark.parametrize("seed", range(12))
def test_randomised_render_and_crop_sweep(seed):
    mesh, K, rgb, depth, poses, hw = _random_case(seed)
    m = FoundationPose(mesh, K)
    m.upload_frame(rgb, depth)
    np.testing.assert_array_equal(m.xyz_map(), fo.depth_to_xyz(depth, K))
    # the sampler on a frame of noise: erosion bit-exact, bilateral filter to 2e-6 relative, translation guess (bounding box of a ragged mask,
    # exact median of the filtered depth under it) bit-exact
    e, bl = m.filter_depth()
    np.testing.assert_array_equal(e, fo.erode_depth(depth))
    np.testing.assert_allclose(bl, fo.bilateral_filter_depth(fo.erode_depth(depth)), rtol=2e-6, atol=0)
    rng = np.random.default_rng(2000 + seed)
    mask = np.zeros(hw, np.uint8)
    y0, x0 = int(rng.integers(0, hw[0] // 2)), int(rng.integers(0, hw[1] // 2))
    mask[y0:y0 + hw[0] // 3, x0:x0 + hw[1] // 3] = (rng.uniform(size=(hw[0] // 3, hw[1] // 3)) < 0.4) * int(rng.integers(1, 256))
    hyp = m.get_hyp_poses(mask)
    ref = fo.get_hyp_poses(depth, mask, K)
    if ref is None:
        assert hyp is None
    else:
        ref = syn.from_colmajor(ref)
        np.testing.assert_allclose(hyp[:, :3, :3], ref[:, :3, :3], atol=1e-6)
        np.testing.assert_array_equal(hyp[:, :3, 3], ref[:, :3, 3])
    om = fo.OracleMesh(mesh)
    for ratio in (1.2, 1.1):
        tri, rast = m.debug_rasterize(mesh.name, poses, ratio)
        a, b = m.render_and_transform(mesh.name, poses, ratio)
        ra, rtri, rrast = fo.render(om, syn.to_colmajor(poses), K, hw, ratio, debug=True)
        rb = fo.crop(rgb, depth, K, syn.to_colmajor(poses), ratio, mesh.diameter)
        np.testing.assert_array_equal(tri, rtri)
        np.testing.assert_allclose(rast, rrast, rtol=0, atol=2e-6)
        np.testing.assert_allclose(a, ra, **F32_TOL)
        np.testing.assert_allclose(b, rb, **F32_TOL)
    m.close()
