"""Seeded synthetic assets for tests and bench: no external data is available.

Mesh   : icosphere (subdivision 4: V=2562, F=5120) deformed to an ellipsoid with semi-axes
         (0.048, 0.032, 0.095) m (mustard-bottle-like, diameter ~0.19 m), spherical UVs, analytic normals.
Texture: 512x512 RGB u8, seeded noise low-passed 8x8 + checker ("textured"), or the reference's
         2x2 (100,100,100) fallback ("untextured", assimp_mesh_loader.cpp:217-222).
Camera : K = [[320,0,320],[0,320,240]] at 640x480 (x2 at 1280x720).
Scene  : object at t=(0.02,-0.01,0.70) m, rotation from seed 1; depth = analytic ray/ellipsoid z-buffer over a
         plane at 1.5 m + N(0,1mm) noise, 2% dropped pixels; rgb = lambert-shaded texture over noise background.

Pure numpy; independent of both the HIP library and the oracle.
"""
from __future__ import annotations

import dataclasses
import numpy as np

SEMI_AXES = (0.048, 0.032, 0.095)


@dataclasses.dataclass
class Mesh:
    name: str
    vertices: np.ndarray      # [V,3] f32, mesh frame (NOT centred; loader semantics)
    normals: np.ndarray       # [V,3] f32
    texcoords: np.ndarray     # [V,2] f32 (u, v) as a loader returns them (v NOT flipped)
    faces: np.ndarray         # [F,3] i32
    texture: np.ndarray       # [TH,TW,3] u8 RGB
    diameter: float = 0.0
    center: np.ndarray | None = None  # AABB centre

    def finalize(self) -> "Mesh":
        v = self.vertices.astype(np.float32)
        if self.center is None:
            self.center = ((v.max(0) + v.min(0)) / 2.0).astype(np.float32)
        if self.diameter <= 0:
            self.diameter = float(mesh_diameter(v))
        return self


def mesh_diameter(v: np.ndarray) -> np.float32:
    """max pairwise vertex distance (assimp_mesh_loader.cpp:47-60), blocked numpy, float32 like the loader."""
    v = v.astype(np.float32)
    best = np.float32(0)
    for i in range(0, len(v), 512):
        d = v[i:i + 512, None, :] - v[None, :, :]
        best = max(best, np.sqrt((d * d).sum(-1, dtype=np.float32)).max())
    return np.float32(best)


def _icosphere(subdiv: int):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
             (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
             (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache = {}
        nf = []

        def mid(i, j):
            key = (min(i, j), max(i, j))
            if key not in cache:
                p = (verts[i] + verts[j]) / 2.0
                verts.append(p / np.linalg.norm(p))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int32)


def make_texture(seed: int = 0, size: int = 512) -> np.ndarray:
    rng = np.random.default_rng(seed)
    noise = rng.uniform(0, 255, size=(size // 8, size // 8, 3))
    img = np.kron(noise, np.ones((8, 8, 1)))
    # cheap low-pass: average with shifted copies
    img = (img + np.roll(img, 4, 0) + np.roll(img, 4, 1) + np.roll(img, (4, 4), (0, 1))) / 4.0
    yy, xx = np.mgrid[0:size, 0:size]
    checker = (((yy // 32) + (xx // 32)) % 2) * 60.0 - 30.0
    img = np.clip(img + checker[..., None], 0, 255)
    return img.astype(np.uint8)


def make_mesh(subdiv: int = 4, textured: bool = True, name: str = "ellipsoid",
              offset=(0.0, 0.0, 0.0)) -> Mesh:
    s, faces = _icosphere(subdiv)
    ax = np.array(SEMI_AXES)
    v = s * ax + np.array(offset)
    n = s / ax
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    u = 0.5 + np.arctan2(s[:, 1], s[:, 0]) / (2 * np.pi)
    w = 0.5 + np.arcsin(np.clip(s[:, 2], -1, 1)) / np.pi
    tex = make_texture() if textured else np.full((2, 2, 3), 100, np.uint8)
    return Mesh(name, v.astype(np.float32), n.astype(np.float32),
                np.stack([u, w], 1).astype(np.float32), faces, tex).finalize()


def intrinsics(W: int = 640, H: int = 480) -> np.ndarray:
    f = 320.0 * W / 640.0
    return np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], dtype=np.float32)


def random_rotation(seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def pose_matrix(R: np.ndarray, t) -> np.ndarray:
    """4x4 row-major numpy pose (use to_colmajor() before handing to the library / oracle)."""
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def to_colmajor(poses: np.ndarray) -> np.ndarray:
    """[..,4,4] numpy (row-major) -> [..,16] column-major float32 as the C ABI expects."""
    p = np.asarray(poses, dtype=np.float32)
    return np.ascontiguousarray(np.swapaxes(p, -1, -2)).reshape(p.shape[:-2] + (16,))


def from_colmajor(flat: np.ndarray) -> np.ndarray:
    f = np.asarray(flat, dtype=np.float32)
    return np.swapaxes(f.reshape(f.shape[:-1] + (4, 4)), -1, -2).copy()


@dataclasses.dataclass
class Scene:
    K: np.ndarray        # [3,3] f32
    rgb: np.ndarray      # [H,W,3] u8
    depth: np.ndarray    # [H,W] f32 metres
    mask: np.ndarray     # [H,W] u8
    gt_pose: np.ndarray  # [4,4] f32: CENTRED mesh -> camera (what Register/Track return)


def make_scene(mesh: Mesh, W: int = 640, H: int = 480, t=(0.02, -0.01, 0.70), rot_seed: int = 1,
               noise_seed: int = 2, drop_seed: int = 3, bg_seed: int = 4, pose=None) -> Scene:
    """pose (4x4 centred-mesh -> camera) overrides t / rot_seed (used for synthetic sequences)"""
    K = intrinsics(W, H)
    R = random_rotation(rot_seed)
    t = np.array(t, dtype=np.float64)
    if pose is not None:
        R, t = np.asarray(pose, np.float64)[:3, :3], np.asarray(pose, np.float64)[:3, 3]
    ax = np.array(SEMI_AXES)
    # rays in camera frame -> object frame (object frame = centred ellipsoid)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    d_cam = np.stack([(xx - K[0, 2]) / K[0, 0], (yy - K[1, 2]) / K[1, 1], np.ones_like(xx)], -1)
    o = R.T @ (-t)
    d = d_cam @ R  # (R^T d)
    oo, dd = o / ax, d / ax
    A = (dd * dd).sum(-1)
    B = 2 * (dd * oo).sum(-1)
    C = (oo * oo).sum() - 1.0
    disc = B * B - 4 * A * C
    hit = disc > 0
    s = np.where(hit, (-B - np.sqrt(np.where(hit, disc, 0))) / (2 * A), 0.0)
    hit &= s > 0
    depth = np.where(hit, s, 1.5)  # d_cam.z == 1 so ray parameter == depth
    p_obj = o + d * s[..., None]
    sph = p_obj / ax
    sph /= np.maximum(np.linalg.norm(sph, axis=-1, keepdims=True), 1e-9)
    u = 0.5 + np.arctan2(sph[..., 1], sph[..., 0]) / (2 * np.pi)
    v = 0.5 + np.arcsin(np.clip(sph[..., 2], -1, 1)) / np.pi
    tex = mesh.texture
    th, tw = tex.shape[:2]
    # the renderer samples texture at (u, 1-v) with row 0 at v'=0
    tx = np.clip((u * tw).astype(int), 0, tw - 1)
    ty = np.clip(((1.0 - v) * th).astype(int), 0, th - 1)
    col = tex[ty, tx].astype(np.float64) / 255.0
    n_obj = p_obj / (ax * ax)
    n_obj /= np.maximum(np.linalg.norm(n_obj, axis=-1, keepdims=True), 1e-12)
    n_cam = n_obj @ R.T
    lam = np.clip(-n_cam[..., 2], 0, 1)
    col = np.clip(col * (0.8 + 0.5 * lam)[..., None], 0, 1)
    rng_bg = np.random.default_rng(bg_seed)
    bg = rng_bg.uniform(0, 1, size=(H, W, 3))
    rgb = np.where(hit[..., None], col, bg)
    rgb = (rgb * 255.0 + 0.5).astype(np.uint8)
    rng_n = np.random.default_rng(noise_seed)
    depth = depth + rng_n.normal(0, 0.001, size=depth.shape)
    rng_d = np.random.default_rng(drop_seed)
    depth = np.where(rng_d.uniform(size=depth.shape) < 0.02, 0.0, depth)
    return Scene(K, rgb, depth.astype(np.float32), (hit * 255).astype(np.uint8),
                 pose_matrix(R, t))


def _family_scene(mesh: Mesh, seed: int, W: int, H: int) -> Scene:
    """one member of the synthetic 'deployment scene family': object 0.55-0.95 m away, up to +-6 cm off axis, any orientation, its own
    depth noise / dropped pixels / background"""
    rng = np.random.default_rng(seed)
    z = rng.uniform(0.55, 0.95)
    t = (rng.uniform(-0.06, 0.06), rng.uniform(-0.05, 0.05), z)
    return make_scene(mesh, W=W, H=H, t=t, rot_seed=seed, noise_seed=seed + 1, drop_seed=seed + 2, bg_seed=seed + 3)


def calibration_scenes(mesh: Mesh, k: int = 8, W: int = 640, H: int = 480):
    """K seeded scenes for fp_calibrate_begin / _add_frame / _finish (seeds 1000, 1010, ...): disjoint from heldout_scenes."""
    return [_family_scene(mesh, 1000 + 10 * i, W, H) for i in range(k)]


def heldout_scenes(mesh: Mesh, k: int = 4, W: int = 640, H: int = 480):
    """scenes no calibration has seen (seeds 5000, 5010, ...; same family, other positions / rotations / noise / background)"""
    return [_family_scene(mesh, 5000 + 10 * i, W, H) for i in range(k)]


def perturb_pose(pose: np.ndarray, deg: float = 5.0, trans: float = 0.01, seed: int = 5) -> np.ndarray:
    rng = np.random.default_rng(seed)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    a = np.deg2rad(deg)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    dR = np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * (Kx @ Kx)
    dt = rng.normal(size=3)
    dt *= trans / np.linalg.norm(dt)
    out = pose.astype(np.float64).copy()
    out[:3, :3] = dR @ out[:3, :3]
    out[:3, 3] += dt
    return out.astype(np.float32)
