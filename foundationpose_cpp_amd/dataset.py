"""RGB-D sequence I/O in the reference's dataset layout (test_data/download.md:6-15)

    <dir>/cam_K.txt   rgb/<id>.png   depth/<id>.png (u16 millimetres)   masks/<id>.png   mesh/<textured .obj>

* `Sequence(dir)` reads it exactly like simple_tests/include/tests/help_func.hpp (ReadCamK :108-129,
  ReadRgbDepthMask :10-36, ReadRgbDepth :38-53) -- decoding goes through the C ABI (fp_read_rgb_depth_mask), so the
  C++ demo (examples/fp_demo.cpp) and Python see identical pixels.
* `write_synthetic_sequence(dir, n_frames)` writes a seeded synthetic sequence in that layout (the mustard data is a
  Google-Drive download, unavailable offline): the SURVEY.md §8d scene with the object moving along a smooth path.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import zlib

import numpy as np

from . import _lib
from . import synthetic as syn


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Sequence:
    def __init__(self, root: str):
        self.root = root
        L = _lib.lib()
        K = np.zeros(9, np.float32)
        if L.fp_read_cam_k(os.path.join(root, "cam_K.txt").encode(), _p(K)):
            raise RuntimeError(_lib.last_error())
        self.K = K.reshape(3, 3)
        # sorted stems of rgb/*.png, like get_files_in_directory + std::sort (test_foundationpose.cpp:73-79)
        self.ids = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(root, "rgb")) if f.endswith(".png"))
        if not self.ids:
            raise RuntimeError(f"no frames under {root}/rgb")
        h, w = C.c_int(), C.c_int()
        if L.fp_frame_size(self._path("rgb", 0).encode(), C.byref(h), C.byref(w)):
            raise RuntimeError(_lib.last_error())
        self.H, self.W = h.value, w.value

    def __len__(self):
        return len(self.ids)

    def _path(self, sub, i):
        return os.path.join(self.root, sub, self.ids[i] + ".png")

    def mesh_path(self) -> str:
        d = os.path.join(self.root, "mesh")
        objs = sorted(f for f in os.listdir(d) if f.endswith(".obj"))
        if not objs:
            raise RuntimeError(f"no .obj under {d}")
        return os.path.join(d, objs[0])

    def frame(self, i: int, with_mask: bool = False):
        """-> rgb u8 [H,W,3], depth f32 [H,W] metres (, mask u8 [H,W])"""
        rgb = np.zeros((self.H, self.W, 3), np.uint8)
        depth = np.zeros((self.H, self.W), np.float32)
        mask = np.zeros((self.H, self.W), np.uint8) if with_mask else None
        rc = _lib.lib().fp_read_rgb_depth_mask(self._path("rgb", i).encode(), self._path("depth", i).encode(),
                                               self._path("masks", i).encode() if with_mask else None, self.H, self.W,
                                               _p(rgb), _p(depth), _p(mask) if with_mask else None)
        if rc:
            raise RuntimeError(_lib.last_error())
        return (rgb, depth, mask) if with_mask else (rgb, depth)


def draw_bbox3d(rgb: np.ndarray, K, pose_bbox_in_cam, dimension) -> np.ndarray:
    """draw3DBoundingBox (help_func.hpp:55-106) on a copy of rgb; pose = ConvertPoseMesh2BBox(pose, mesh)."""
    out = np.ascontiguousarray(rgb, np.uint8).copy()
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    p = np.ascontiguousarray(np.asarray(pose_bbox_in_cam, np.float32).T).reshape(16)   # column-major
    d = np.ascontiguousarray(dimension, np.float32)
    if _lib.lib().fp_draw_bbox3d(_p(out), out.shape[0], out.shape[1], _p(K), _p(p), _p(d)):
        raise RuntimeError(_lib.last_error())
    return out


def convert_pose_mesh2bbox(pose_in_mesh: np.ndarray, mesh) -> np.ndarray:
    """ConvertPoseMesh2BBox (mesh_loader.hpp:75-81): pose * T(-centre) * orient_bounds."""
    tf = np.eye(4, dtype=np.float32)
    tf[:3, 3] = -np.asarray(mesh.center, np.float32)
    return np.asarray(pose_in_mesh, np.float32) @ tf @ np.asarray(mesh.orient_bounds, np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# writer side (synthetic data only)


def write_png(path: str, img: np.ndarray) -> None:
    """8-bit grey / RGB or 16-bit grey PNG (what the dataset layout uses), stdlib zlib only."""
    a = np.ascontiguousarray(img)
    if a.dtype == np.uint16 and a.ndim == 2:
        ctype, depth, rows = 0, 16, a.astype(">u2").tobytes()
        stride = a.shape[1] * 2
    elif a.dtype == np.uint8 and a.ndim == 2:
        ctype, depth, rows, stride = 0, 8, a.tobytes(), a.shape[1]
    elif a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3:
        ctype, depth, rows, stride = 2, 8, a.tobytes(), a.shape[1] * 3
    else:
        raise ValueError("write_png: u8 [H,W], u8 [H,W,3] or u16 [H,W]")
    H, Wd = a.shape[:2]
    raw = b"".join(b"\x00" + rows[y * stride:(y + 1) * stride] for y in range(H))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", Wd, H, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_obj(d: str, mesh, name: str = "textured.obj") -> str:
    """mesh/ directory: OBJ + MTL + texture PNG in the form fp_mesh_load_obj (and assimp) read"""
    os.makedirs(d, exist_ok=True)
    write_png(os.path.join(d, "texture_map.png"), mesh.texture)
    with open(os.path.join(d, "material.mtl"), "w") as f:
        f.write("newmtl material_0\nKd 1 1 1\nmap_Kd texture_map.png\n")
    with open(os.path.join(d, name), "w") as f:
        f.write("mtllib material.mtl\no object\n")
        f.writelines("v %.9g %.9g %.9g\n" % tuple(p) for p in mesh.vertices)
        f.writelines("vt %.9g %.9g\n" % tuple(p) for p in mesh.texcoords)
        f.writelines("vn %.9g %.9g %.9g\n" % tuple(p) for p in mesh.normals)
        f.write("usemtl material_0\n")
        f.writelines("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c, c, c) for a, b, c in mesh.faces + 1)
    return os.path.join(d, name)


def write_synthetic_sequence(root: str, n_frames: int = 8, width: int = 640, height: int = 480, mesh=None):
    """-> (mesh, [ground-truth centred-mesh->camera poses]).  Frame ids are zero-padded like the mustard data."""
    mesh = mesh or syn.make_mesh()
    for sub in ("rgb", "depth", "masks"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    write_obj(os.path.join(root, "mesh"), mesh)
    K = syn.intrinsics(width, height)
    with open(os.path.join(root, "cam_K.txt"), "w") as f:
        f.writelines(" ".join("%.18e" % v for v in row) + "\n" for row in np.asarray(K, np.float64))
    base = syn.make_scene(mesh, width, height)
    gts = []
    for i in range(n_frames):
        # smooth motion: 2 degrees about a fixed axis and 3 mm sideways per frame
        ang = np.radians(2.0 * i)
        ax = np.array([0.3, 1.0, 0.2]) / np.linalg.norm([0.3, 1.0, 0.2])
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        dR = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        pose = base.gt_pose.copy()
        pose[:3, :3] = (dR @ base.gt_pose[:3, :3]).astype(np.float32)
        pose[:3, 3] = base.gt_pose[:3, 3] + np.array([0.003 * i, -0.001 * i, 0.002 * i], np.float32)
        sc = syn.make_scene(mesh, width, height, pose=pose)
        fid = "%07d" % (1581120424100262102 % 10 ** 7 + i)
        write_png(os.path.join(root, "rgb", fid + ".png"), sc.rgb)
        write_png(os.path.join(root, "depth", fid + ".png"), np.clip(np.rint(sc.depth * 1000.0), 0, 65535).astype(np.uint16))
        write_png(os.path.join(root, "masks", fid + ".png"), (sc.mask > 0).astype(np.uint8) * 255)
        gts.append(pose)
    return mesh, gts
