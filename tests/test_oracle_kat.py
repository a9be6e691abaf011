"""Known-answer tests that pin the CPU oracle (oracle/fp_oracle.c).

The reference holds no golden vectors for this path, so every expected value below is
derived by hand from the formulas of the cited reference lines.  Runs on CPU.
"""
import numpy as np
import pytest

from foundationpose_cpp_amd import synthetic as syn
from oracle import fp_oracle as fo


def _quad_mesh(half=0.05, z=0.0, diameter=0.2, tex_val=200):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], np.float32)
    n = np.tile(np.array([[0, 0, -1]], np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    tex = np.full((2, 2, 3), tex_val, np.uint8)
    return syn.Mesh("quad", v, n, uv, f, tex, diameter=diameter, center=np.zeros(3, np.float32))


K = syn.intrinsics()


@pytest.fixture
def float_model(request):
    """runs a KAT under the float model named by its `fmad` parameter (fp_oracle.c: contracted like nvcc -fmad=true /
    separately rounded) and restores the default afterwards.  The KAT inputs are small integers / powers of two, so
    every product and sum is exact and BOTH models must reproduce the hand-derived answer."""
    fo.set_fmad(request.getfixturevalue("fmad"))
    yield
    fo.set_fmad(True)


def test_rotation_grid_counts_and_structure():
    # foundationpose_sampling.cpp:100 subdivides once (12 -> 42 >= 40); 42 views x 6 in-plane = 252 (:219)
    ico = fo.icosphere()
    assert ico.shape == (42, 3)
    t = (1 + 5 ** 0.5) / 2
    np.testing.assert_allclose(ico[0], np.array([-1, t, 0]) / np.sqrt(1 + t * t), atol=1e-7)
    np.testing.assert_allclose(np.linalg.norm(ico, axis=1), 1, atol=1e-6)
    # vertex 12 is the midpoint of edge (0,11) of the first face, normalised
    mid = (ico[0] + ico[11]) / 2
    np.testing.assert_allclose(ico[12], mid / np.linalg.norm(mid), atol=1e-6)
    g = syn.from_colmajor(fo.rotation_grid())
    assert g.shape == (252, 4, 4)
    R = g[:, :3, :3]
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3), (252, 1, 1)), atol=1e-6)
    np.testing.assert_allclose(np.linalg.det(R), 1, atol=1e-6)
    # cam_in_ob = inverse: camera position = view vertex, z axis = -vertex (:185-188)
    cam = np.linalg.inv(g.astype(np.float64))
    np.testing.assert_allclose(cam[::6, :3, 3], ico, atol=1e-6)
    np.testing.assert_allclose(cam[::6, :3, 2], -ico, atol=1e-6)
    # in-plane minor order: pose[6i+k] = Rz(60k deg)^-1 * pose[6i]
    for k in range(6):
        a = np.deg2rad(60 * k)
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        np.testing.assert_allclose(g[6 * 7 + k, :3, :3], Rz.T @ g[6 * 7, :3, :3], atol=1e-6)
    assert fo.rotation_grid(40, 15).shape[0] == 1008
    # all 252 rotations are distinct (ClusterPoses result is discarded, :235)
    flat = np.round(R.reshape(252, 9), 4)
    assert len(np.unique(flat, axis=0)) == 252


def test_crop_window_tf_hand_pose():
    # t=(0,0,1), f=320, c=(320,240), diam=0.2, ratio=1.2 -> r=0.12, radius=fy*r/z=38.4
    # left=round(281.6)=282 right=round(358.4)=358 top=round(201.6)=202 bottom=round(278.4)=278
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = 1.0
    tf = fo.crop_window_tf(syn.to_colmajor(pose[None]), K, 1.2, 0.2)[0].reshape(3, 3)
    s = np.float32(160) / np.float32(76)
    np.testing.assert_allclose(tf, [[s, 0, -282 * s], [0, s, -202 * s], [0, 0, 1]], rtol=1e-6)
    bb = fo.bbox2d(tf.reshape(1, 9))[0]
    np.testing.assert_allclose(bb, [282, 202, 282 + 159 / s, 202 + 159 / s], rtol=1e-6)
    # only the translation matters, and only the v column for the radius (foundationpose_render.cpp:59-66)
    pose2 = pose.copy()
    pose2[:3, :3] = syn.random_rotation(3)
    tf2 = fo.crop_window_tf(syn.to_colmajor(pose2[None]), K, 1.2, 0.2)[0].reshape(3, 3)
    np.testing.assert_array_equal(tf, tf2)


def test_projection_matrix():
    P = fo.projection_matrix(K, 480, 640).reshape(4, 4).T
    exp = np.array([[1, 0, 0, 0], [0, 640 / 480, 0, 0], [0, 0, -(100.1) / 99.9, -2 * 10 / 99.9], [0, 0, -1, 0]])
    np.testing.assert_allclose(P, exp, rtol=1e-6, atol=1e-7)


def test_depth_to_xyz_and_filters():
    d = np.full((480, 640), 0.5, np.float32)
    d[0, 0] = 0.0
    xyz = fo.depth_to_xyz(d, K)
    np.testing.assert_array_equal(xyz[0, 0], 0)
    np.testing.assert_allclose(xyz[240, 320], [0, 0, 0.5])
    np.testing.assert_allclose(xyz[100, 600], [(600 - 320) * 0.5 / 320, (100 - 240) * 0.5 / 320, 0.5], rtol=1e-6)
    e = fo.erode_depth(d)
    assert e[0, 0] == 0 and e[5, 5] == 0.5 and e[1, 1] == 0.5   # 1 bad of 9..25 neighbours << 0.8
    b = fo.bilateral_filter_depth(e)
    np.testing.assert_allclose(b[5, 5], 0.5, rtol=1e-6)
    np.testing.assert_allclose(b[0, 0], 0.5, rtol=1e-6)          # invalid centre is filled from valid neighbours
    # erode: isolated pixel surrounded by invalid -> ratio 24/25 > 0.8 -> 0
    d2 = np.zeros((20, 20), np.float32)
    d2[10, 10] = 0.5
    assert fo.erode_depth(d2)[10, 10] == 0
    # a depth step of 2 mm > 1 mm threshold counts as bad: column edge pixel has 10/25 bad -> kept
    d3 = np.full((20, 20), 0.5, np.float32)
    d3[:, 10:] = 0.502
    e3 = fo.erode_depth(d3)
    assert e3[10, 9] == 0.5 and e3[10, 10] == np.float32(0.502)


def test_guess_translation():
    d = np.full((480, 640), 0.8, np.float32)
    m = np.zeros((480, 640), np.uint8)
    m[100:201, 300:401] = 255          # u in [300,400], v in [100,200] -> uc=350, vc=150
    c = fo.guess_translation(d, m, K)
    np.testing.assert_allclose(c, [(350 - 320) / 320 * 0.8, (150 - 240) / 320 * 0.8, 0.8], rtol=3e-6)
    # even count -> mean of the two middle values (foundationpose_sampling.cpp:293-294)
    d2 = d.copy()
    m2 = np.zeros_like(m)
    m2[10, 10:14] = 1
    d2[10, 10:14] = [0.1, 0.2, 0.4, 0.9]
    c2 = fo.guess_translation(d2, m2, K)
    np.testing.assert_allclose(c2[2], 0.3, rtol=1e-6)
    assert fo.guess_translation(d, np.zeros_like(m), K) is None            # empty mask -> false (:269)
    assert fo.guess_translation(np.zeros_like(d), m, K) is None            # no valid depth -> false (:278)


def test_refine_post_process_closed_forms():
    pose = syn.pose_matrix(syn.random_rotation(9), (0.1, -0.2, 0.7))
    p16 = syn.to_colmajor(pose[None])
    z = np.zeros((1, 3), np.float32)
    np.testing.assert_allclose(fo.refine_post_process(p16, z, z, 0.2), p16, atol=1e-7)  # zero update = identity
    # trans scaled by diam/2 and added in the camera frame (foundationpose.cpp:384-385,397)
    out = syn.from_colmajor(fo.refine_post_process(p16, np.array([[1, 2, 3]], np.float32), z, 0.2))[0]
    np.testing.assert_allclose(out[:3, 3], pose[:3, 3] + 0.1 * np.array([1, 2, 3]), rtol=1e-6, atol=1e-7)
    # rot=(atanh(1/2),0,0) -> v=(0.5*0.349..,0,0): Rdelta = Rx(angle)^T, left-multiplied (:388-400)
    ang = 0.5 * 0.349065850398865
    rot = np.array([[np.arctanh(0.5), 0, 0]], np.float32)
    out = syn.from_colmajor(fo.refine_post_process(p16, z, rot, 0.2))[0]
    Rx = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
    np.testing.assert_allclose(out[:3, :3], Rx.T @ pose[:3, :3], atol=1e-6)


def test_argmax_first_max():
    assert fo.argmax(np.array([0.1, 0.9, 0.9, 0.3], np.float32)) == 1
    assert fo.argmax(np.array([-5.0], np.float32)) == 0


@pytest.mark.parametrize("fmad", [False, True])
def test_render_quad_geometry_and_fill_rule(fmad, float_model):
    """A 0.1 m fronto-parallel square at z=1 m spans image u in [304,336], v in [224,256] (GL projection:
    ndc = 2u/W-1, i.e. u is a continuous coordinate with pixel i covering [i,i+1]).
    generate_pose_clip maps u in [bbox.l, bbox.r] = [282, 282+159/s] (s=160/76; ConstructBBox2D uses W-1=159,
    foundationpose_render.cpp:126) onto the 160 raster pixels, so raster scale g = 160*s/159 and crop pixel
    centre c+0.5 <-> u = 282 + (c+0.5)/g.  (u-282)*g in [46.61,114.40] -> c in [47,113]: 67 px."""
    mesh = _quad_mesh()
    om = fo.OracleMesh(mesh)
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = 1.0
    out, tri, rast = fo.render(om, syn.to_colmajor(pose[None]), K, (480, 640), 1.2, debug=True)
    cov = tri[0] > 0
    ys, xs = np.nonzero(cov)
    # tri is in raster (y-up) order: output rows 47..113 are raster rows 159-113..159-47 = 46..112
    assert xs.min() == 47 and xs.max() == 113 and ys.min() == 46 and ys.max() == 112
    assert cov.sum() == 67 * 67                       # shared diagonal rasterised exactly once, no holes
    assert set(np.unique(tri[0])) == {0, 1, 2}
    # raster is y-up: triangle 1 = (v0,v1,v2) holds camera-frame (+x,-y..+y) half -> check via flip in output
    o = out[0]
    fg = (o[..., 3:] != 0).any(-1) | (o[..., :3] != 0).any(-1)
    assert fg.sum() == 67 * 67
    oy, ox = np.nonzero(fg)
    assert oy.min() == 47 and oy.max() == 113       # vertical flip applied (foundationpose_render.cpp:676-680)
    # xyz channel = (p - t)/(diam/2): at crop pixel (x=100,y=60): image px = 282 + (100+0.5)/s - 0.5
    g = 160 * (160 / 76) / 159
    u = 282 + (100 + 0.5) / g
    v = 202 + (60 + 0.5) / g
    X = (u - 320) / 320 * 1.0
    Y = (v - 240) / 320 * 1.0
    np.testing.assert_allclose(o[60, 100, 3:], [X / 0.1, Y / 0.1, 0], atol=2e-5)
    # colour: texture 200/255 * (0.8 + 0.5*diffuse), diffuse = clamp(-n_cam.z) = 1 -> clamp(1.0196) = 1
    np.testing.assert_allclose(o[60, 100, :3], min(1.0, 200 / 255 * 1.3), atol=1e-6)
    # rast_out: barycentrics within [0,1], z/w = clip z/w at 1 m: (-(100.1/99.9)*(-1) - 20/99.9... ) evaluated below
    zw = ((100.1 / 99.9) * 1.0 - 2 * 10 / 99.9) / 1.0
    np.testing.assert_allclose(rast[0][cov][:, 2], zw, rtol=1e-5)
    assert (rast[0][cov][:, :2] >= 0).all() and (rast[0][cov][:, :2] <= 1).all()


@pytest.mark.parametrize("fmad", [False, True])
def test_render_depth_test_nearest_wins_and_tie_rule(fmad, float_model):
    # two coincident quads (4 triangles): equal depth everywhere -> the later triangles win (FineRaster ROP rule)
    q = _quad_mesh()
    v = np.concatenate([q.vertices, q.vertices])
    f = np.concatenate([q.faces, q.faces + 4])
    mesh = syn.Mesh("dq", v, np.concatenate([q.normals] * 2), np.concatenate([q.texcoords] * 2), f, q.texture,
                    diameter=0.2, center=np.zeros(3, np.float32))
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = 1.0
    _, tri, _ = fo.render(fo.OracleMesh(mesh), syn.to_colmajor(pose[None]), K, (480, 640), 1.2, debug=True)
    assert set(np.unique(tri[0])) == {0, 3, 4}
    # second quad 1 cm nearer to the camera, listed FIRST -> still wins on depth
    v2 = v.copy()
    v2[:4, 2] -= 0.01
    mesh2 = syn.Mesh("dq2", v2, mesh.normals, mesh.texcoords, f, q.texture, diameter=0.2, center=np.zeros(3, np.float32))
    _, tri2, _ = fo.render(fo.OracleMesh(mesh2), syn.to_colmajor(pose[None]), K, (480, 640), 1.2, debug=True)
    inner = tri2[0][60:100, 60:100]
    assert set(np.unique(inner)) <= {1, 2}


@pytest.mark.parametrize("fmad", [False, True])
def test_render_frustum_clip_path(fmad, float_model):
    # a huge triangle crossing the near plane exercises clipTriangleWithFrustum; coverage must stay bounded and sane
    v = np.array([[-5, -5, 2.0], [5, -5, 2.0], [0, 5, -3.0]], np.float32)   # third vertex behind the camera
    mesh = syn.Mesh("big", v, np.tile(np.array([[0, 0, -1]], np.float32), (3, 1)), np.zeros((3, 2), np.float32),
                    np.array([[0, 1, 2]], np.int32), np.full((2, 2, 3), 100, np.uint8), diameter=0.2,
                    center=np.zeros(3, np.float32))
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = 1.0
    out, tri, rast = fo.render(fo.OracleMesh(mesh), syn.to_colmajor(pose[None]), K, (480, 640), 1.2, debug=True)
    assert np.isfinite(out).all()
    assert (tri[0] > 0).sum() > 1000          # the visible part fills a good part of the crop
    assert set(np.unique(tri[0])) <= {0, 1}


def test_crop_matches_hand_sampling():
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    depth = rng.uniform(0.5, 1.5, size=(480, 640)).astype(np.float32)
    depth[200:210, 300:310] = 0
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = 1.0
    out = fo.crop(rgb, depth, K, syn.to_colmajor(pose[None]), 1.2, 0.2)[0]
    s = np.float32(160) / np.float32(76)
    # dst (x=0,y=0) samples src (282,202) exactly (integer coordinates = pixel centres)
    np.testing.assert_allclose(out[0, 0, :3], rgb[202, 282] / 255.0, atol=1e-6)
    d = depth[202, 282]
    exp = np.array([(282 - 320) * d / 320, (202 - 240) * d / 320, d - 1.0]) / 0.1
    exp = np.where(np.abs(exp) > 4, 0, exp)
    np.testing.assert_allclose(out[0, 0, 3:], exp, rtol=1e-5, atol=1e-6)
    # bilinear in u8 then /255 -> every value is k/255
    k = out[..., :3] * 255
    np.testing.assert_allclose(k, np.round(k), atol=1e-4)
    # invalid depth -> xyz 0 for the whole pixel
    x = int(round((305 - 282) * s)); y = int(round((205 - 202) * s))
    np.testing.assert_array_equal(out[y, x, 3:], 0)
    # window outside the image -> zeros
    pose2 = pose.copy()
    pose2[0, 3] = 5.0
    out2 = fo.crop(rgb, depth, K, syn.to_colmajor(pose2[None]), 1.2, 0.2)[0]
    np.testing.assert_array_equal(out2, 0)


def test_mesh_stats(syn_mesh):
    assert syn_mesh.vertices.shape == (2562, 3) and syn_mesh.faces.shape == (5120, 3)
    np.testing.assert_allclose(fo.mesh_diameter(syn_mesh.vertices), 0.19, rtol=1e-5)
    np.testing.assert_allclose(fo.mesh_center(syn_mesh.vertices + 0.5), [0.5, 0.5, 0.5], atol=1e-6)


def test_float_models_differ_only_in_rounding(syn_mesh, syn_scene):
    """contracted vs separately rounded rendering of three hypotheses of the synthetic scene: identical triangle ids (the
    integer part of the pipeline sees the same snapped vertices), tensors equal up to a texel-boundary flip"""
    om = fo.OracleMesh(syn_mesh)
    poses = fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, syn_scene.K)[[0, 100, 251]]
    out = {}
    try:
        for fm in (False, True):
            fo.set_fmad(fm)
            out[fm] = fo.render(om, poses, syn_scene.K, syn_scene.depth.shape, 1.2, debug=True)
    finally:
        fo.set_fmad(True)
    np.testing.assert_array_equal(out[False][1], out[True][1])
    d = np.abs(out[False][0] - out[True][0])
    assert 0 < d.max() < 2e-2 and d.mean() < 1e-6, (d.max(), d.mean())     # the models are really different, and only slightly


@pytest.mark.parametrize("ratio", [1.2, 1.1])
@pytest.mark.parametrize("pose_seed", [1, 11])
def test_render_of_an_ellipsoid_matches_the_analytic_surface(ratio, pose_seed):
    """An INDEPENDENT pin of the whole rendering chain (crop window -> bbox remap -> projection -> raster at pixel centres ->
    interpolation -> Lambert shading -> vertical flips -> xyz normalisation): the oracle's render of a finely tessellated
    ellipsoid against the closed-form ray / ellipsoid intersection.  What had to be learned from it is the reference's sampling
    convention: crop pixel x looks at image coordinate b0 + (x + 0.5) (b2 - b0) / 160 with (b0, b2) = tf^-1 (0, 159)
    (ConstructBBox2D uses W - 1, foundationpose_render.cpp:123-149) -- NOT tf^-1 (x), so the rendered crop is 0.6 % narrower than
    the observed one; with any other convention the error below is 50-100x larger."""
    mesh = syn.make_mesh(subdiv=5, textured=False)
    R = syn.random_rotation(pose_seed)
    t = np.array([0.02, -0.01, 0.70]) if pose_seed == 1 else np.array([-0.11, 0.07, 0.55])
    P = syn.pose_matrix(R, t)
    K = syn.intrinsics().astype(np.float64)
    p16 = syn.to_colmajor(P[None])
    out = fo.render(fo.OracleMesh(mesh), p16, syn.intrinsics(), (480, 640), ratio)[0]
    tf = fo.crop_window_tf(p16, syn.intrinsics(), ratio, mesh.diameter)[0].reshape(3, 3).astype(np.float64)
    inv = np.linalg.inv(tf)
    b0x, b0y, b2x, b2y = inv[0, 2], inv[1, 2], inv[0, 0] * 159 + inv[0, 2], inv[1, 1] * 159 + inv[1, 2]
    yy, xx = np.mgrid[0:160, 0:160].astype(np.float64)
    sx, sy = b0x + (xx + 0.5) * (b2x - b0x) / 160, b0y + (yy + 0.5) * (b2y - b0y) / 160
    d_cam = np.stack([(sx - K[0, 2]) / K[0, 0], (sy - K[1, 2]) / K[1, 1], np.ones_like(sx)], -1)
    ax = np.array(syn.SEMI_AXES)
    o, d = R.T @ (-t), d_cam @ R
    oo, dd = o / ax, d / ax
    A, B, C = (dd * dd).sum(-1), 2 * (dd * oo).sum(-1), (oo * oo).sum() - 1
    disc = B * B - 4 * A * C
    hit = disc > 0
    s = np.where(hit, (-B - np.sqrt(np.where(hit, disc, 0))) / (2 * A), 0)
    xyz = np.where(hit[..., None], (d_cam * s[..., None] - t) / (mesh.diameter / 2), 0)
    fg = np.abs(out[..., 3:6]).sum(-1) > 0
    both = fg & hit
    assert (fg & hit).sum() / (fg | hit).sum() > 0.995                       # silhouette (the polyhedron is inscribed)
    inner = both & np.roll(both, 1, 0) & np.roll(both, -1, 0) & np.roll(both, 1, 1) & np.roll(both, -1, 1)
    err = np.abs(out[..., 3:6] - xyz)[inner]
    # subdivision 5: facets ~4 mm, sagitta ~0.05 mm = 5e-4 normalised at grazing angles; mean an order below
    assert err.mean() < 1.5e-4 and np.percentile(err, 99) < 2e-3, (err.mean(), np.percentile(err, 99))
    # Lambert term of the untextured mesh: (100 / 255) (0.8 + 0.5 clamp(-n_cam.z)) with the analytic normal
    p_obj = o + d * s[..., None]
    n_obj = p_obj / (ax * ax)
    n_obj /= np.maximum(np.linalg.norm(n_obj, axis=-1, keepdims=True), 1e-12)
    lam = np.clip(-(n_obj @ R.T)[..., 2], 0, 1)
    want = np.clip(100.0 / 255.0 * (0.8 + 0.5 * lam), 0, 1)
    for c in range(3):
        e = np.abs(out[..., c] - want)[inner]
        assert e.mean() < 5e-4 and np.percentile(e, 99) < 5e-3, (c, e.mean(), np.percentile(e, 99))
