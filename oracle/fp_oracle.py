"""ctypes wrapper over oracle/libfp_oracle.so -- TEST INFRASTRUCTURE ONLY (see fp_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libfp_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("fp_oracle.c", "fp_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.fpo_mesh_diameter.restype = C.c_float
        _LIB.fpo_set_fmad(1)   # default float model = the product's default (contracted like nvcc -fmad=true)
    return _LIB


def set_fmad(on: bool) -> None:
    """Float model of the rendering stage (fp_oracle.c): True = contracted multiply-adds (default), False = separate roundings."""
    lib().fpo_set_fmad(1 if on else 0)


def get_fmad() -> bool:
    return bool(lib().fpo_get_fmad())


class _Mesh(C.Structure):
    _fields_ = [("V", C.c_int), ("F", C.c_int), ("verts", C.c_void_p), ("normals", C.c_void_p),
                ("uvs", C.c_void_p), ("faces", C.c_void_p), ("tex", C.c_void_p), ("TH", C.c_int),
                ("TW", C.c_int), ("diameter", C.c_float)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleMesh:
    """Holds the renderer-side view of a mesh: centred vertices, (u,1-v) uvs (foundationpose_render.cpp:396-406)."""

    def __init__(self, mesh):
        self.verts = _f32(mesh.vertices - mesh.center[None, :].astype(np.float32))
        self.normals = _f32(mesh.normals)
        uv = np.asarray(mesh.texcoords, dtype=np.float32)
        self.uvs = _f32(np.stack([uv[:, 0], np.float32(1) - uv[:, 1]], 1))
        self.faces = np.ascontiguousarray(mesh.faces, dtype=np.int32)
        self.tex = np.ascontiguousarray(mesh.texture, dtype=np.uint8)
        self.diameter = float(mesh.diameter)
        self.c = _Mesh(len(self.verts), len(self.faces), _p(self.verts), _p(self.normals), _p(self.uvs),
                       _p(self.faces), _p(self.tex), self.tex.shape[0], self.tex.shape[1], self.diameter)


def rotation_grid(min_views=40, inplane_step=60):
    cap = 4096
    out = np.zeros((cap, 16), np.float32)
    n = lib().fpo_rotation_grid(min_views, inplane_step, _p(out), cap)
    return out[:n].copy()


def icosphere(min_views=40):
    out = np.zeros((4096, 3), np.float32)
    n = lib().fpo_icosphere(min_views, _p(out), 4096)
    return out[:n].copy()


def depth_to_xyz(depth, K, min_depth=0.001):
    d = _f32(depth)
    H, W = d.shape
    out = np.zeros((H, W, 3), np.float32)
    lib().fpo_depth_to_xyz(_p(d), H, W, C.c_float(K[0, 0]), C.c_float(K[1, 1]), C.c_float(K[0, 2]),
                           C.c_float(K[1, 2]), C.c_float(min_depth), _p(out))
    return out


def erode_depth(depth, radius=2, diff=0.001, ratio=0.8, zfar=100.0):
    d = _f32(depth)
    H, W = d.shape
    out = np.zeros_like(d)
    lib().fpo_erode_depth(_p(d), _p(out), H, W, radius, C.c_float(diff), C.c_float(ratio), C.c_float(zfar))
    return out


def bilateral_filter_depth(depth, zfar=100.0, radius=2, sigmaD=2.0, sigmaR=100000.0):
    d = _f32(depth)
    H, W = d.shape
    out = np.zeros_like(d)
    lib().fpo_bilateral_filter_depth(_p(d), _p(out), H, W, C.c_float(zfar), radius, C.c_float(sigmaD),
                                     C.c_float(sigmaR))
    return out


def guess_translation(depth, mask, K, min_depth=0.001):
    d = _f32(depth)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    Kf = _f32(K)
    c = np.zeros(3, np.float32)
    ok = lib().fpo_guess_translation(_p(d), _p(m), d.shape[0], d.shape[1], _p(Kf), C.c_float(min_depth), _p(c))
    return (c if ok else None)


def get_hyp_poses(depth, mask, K, inplane_step=60):
    d = _f32(depth)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    Kf = _f32(K)
    out = np.zeros((4096, 16), np.float32)
    n = lib().fpo_get_hyp_poses(_p(d), _p(m), d.shape[0], d.shape[1], _p(Kf), inplane_step, _p(out), 4096)
    return out[:n].copy() if n > 0 else None


def crop_window_tf(poses16, K, crop_ratio, diameter, out_hw=(160, 160)):
    p = _f32(poses16)
    Kf = _f32(K)
    tfs = np.zeros((len(p), 9), np.float32)
    lib().fpo_compute_crop_window_tf(_p(p), len(p), _p(Kf), out_hw[0], out_hw[1], C.c_float(crop_ratio),
                                     C.c_float(diameter), _p(tfs))
    return tfs


def bbox2d(tfs, out_hw=(160, 160)):
    t = _f32(tfs)
    out = np.zeros((len(t), 4), np.float32)
    lib().fpo_construct_bbox2d(_p(t), len(t), out_hw[0], out_hw[1], _p(out))
    return out


def projection_matrix(K, H, W, znear=0.1, zfar=100.0):
    Kf = _f32(K)
    P = np.zeros(16, np.float32)
    lib().fpo_projection_matrix(_p(Kf), H, W, C.c_float(znear), C.c_float(zfar), _p(P))
    return P


def render(omesh: OracleMesh, poses16, K, img_hw, crop_ratio, out_hw=(160, 160), min_depth=0.001,
           max_depth=4.0, debug=False):
    p = _f32(poses16)
    Kf = _f32(K)
    N = len(p)
    out = np.zeros((N, out_hw[0], out_hw[1], 6), np.float32)
    tri = np.zeros((N, out_hw[0], out_hw[1]), np.int32) if debug else None
    rast = np.zeros((N, out_hw[0], out_hw[1], 4), np.float32) if debug else None
    lib().fpo_render(C.byref(omesh.c), _p(p), N, _p(Kf), img_hw[0], img_hw[1], out_hw[0], out_hw[1],
                     C.c_float(crop_ratio), C.c_float(min_depth), C.c_float(max_depth), _p(out),
                     _p(tri) if debug else None, _p(rast) if debug else None)
    return (out, tri, rast) if debug else out


def crop(rgb, depth, K, poses16, crop_ratio, diameter, out_hw=(160, 160), min_depth=0.001, max_depth=4.0):
    r = np.ascontiguousarray(rgb, dtype=np.uint8)
    d = _f32(depth)
    p = _f32(poses16)
    Kf = _f32(K)
    N = len(p)
    out = np.zeros((N, out_hw[0], out_hw[1], 6), np.float32)
    lib().fpo_crop(_p(r), _p(d), d.shape[0], d.shape[1], _p(Kf), _p(p), N, out_hw[0], out_hw[1],
                   C.c_float(crop_ratio), C.c_float(diameter), C.c_float(min_depth), C.c_float(max_depth), _p(out))
    return out


def refine_post_process(poses16, trans, rot, diameter):
    p = _f32(poses16)
    t = _f32(trans)
    r = _f32(rot)
    out = np.zeros_like(p)
    lib().fpo_refine_post_process(_p(p), _p(t), _p(r), len(p), C.c_float(diameter), _p(out))
    return out


def argmax(scores):
    s = _f32(scores).ravel()
    return int(lib().fpo_argmax(_p(s), len(s)))


def mesh_diameter(verts):
    v = _f32(verts)
    return float(lib().fpo_mesh_diameter(_p(v), len(v)))


def mesh_center(verts):
    v = _f32(verts)
    c = np.zeros(3, np.float32)
    lib().fpo_mesh_center(_p(v), len(v), _p(c))
    return c


def num_threads():
    return int(lib().fpo_num_threads())
