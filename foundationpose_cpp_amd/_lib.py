"""ctypes loader for libfoundationpose_amd.so (the C ABI of include/foundationpose_amd.h).

The library is the product: there is NO CPU fallback.  If it is missing or cannot be loaded this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FP_LIB_PATH") or os.path.join(_HERE, "libfoundationpose_amd.so")  # FP_LIB_PATH: experimental builds (tools/)
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/foundationpose_amd.h declares
SYMBOLS = [
    "fp_create", "fp_create_on", "fp_device", "fp_register_sharded", "fp_destroy", "fp_last_error", "fp_set_inplane_steps", "fp_num_hypotheses",
    "fp_register", "fp_track", "fp_register_ex", "fp_track_ex", "fp_track_submit", "fp_track_wait", "fp_track_multi",
    "fp_upload_frame", "fp_get_xyz_map", "fp_get_hyp_poses", "fp_filter_depth",
    "fp_render_and_transform", "fp_debug_rasterize", "fp_refiner_infer", "fp_scorer_infer",
    "fp_refine_post_process", "fp_argmax", "fp_register_shard_begin", "fp_register_shard_finish", "fp_register_shard_begin_packed", "fp_register_shard_finish_packed", "fp_download",
    "fp_profile_enable", "fp_profile_reset", "fp_profile_report", "fp_stream", "fp_synchronize",
    "fp_mesh_load_obj", "fp_mesh_free", "fp_mesh_view", "fp_mesh_orient_bounds",
    "fp_image_read_png", "fp_frame_size", "fp_read_rgb_depth_mask", "fp_read_cam_k", "fp_image_write_png_rgb",
    "fp_draw_bbox3d", "fp_set_precision", "fp_get_precision", "fp_calibrate_fp8", "fp_calibrate", "fp_calibrate_begin", "fp_calibrate_add_frame", "fp_calibrate_finish", "fp_calibrate_abort", "fp_calibrate_frames", "fp_calibration_size", "fp_get_calibration_blob", "fp_set_calibration_blob", "fp_get_calibration", "fp_set_calibration", "fp_set_float_model", "fp_get_float_model", "fp_net_create", "fp_net_destroy", "fp_net_max_batch", "fp_net_blob", "fp_net_infer",
]


class FpMesh(C.Structure):
    _fields_ = [("name", C.c_char_p), ("num_vertices", C.c_int), ("num_faces", C.c_int),
                ("vertices", C.c_void_p), ("normals", C.c_void_p), ("texcoords", C.c_void_p),
                ("faces", C.c_void_p), ("texture", C.c_void_p), ("tex_height", C.c_int),
                ("tex_width", C.c_int), ("diameter", C.c_float), ("center", C.c_float * 3)]


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-8000:])
    if res.returncode != 0:
        raise RuntimeError("building libfoundationpose_amd.so failed")
    return LIB_PATH


_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        getattr(L, s)  # raises AttributeError if the library does not export the symbol
    L.fp_create.restype = C.c_void_p
    L.fp_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.fp_create_on.restype = C.c_void_p
    L.fp_create_on.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.fp_destroy.argtypes = [C.c_void_p]
    L.fp_destroy.restype = None
    L.fp_last_error.restype = C.c_char_p
    L.fp_stream.restype = C.c_void_p
    L.fp_stream.argtypes = [C.c_void_p]
    L.fp_mesh_load_obj.restype = C.c_void_p
    L.fp_mesh_load_obj.argtypes = [C.c_char_p, C.c_char_p]
    L.fp_mesh_free.restype = None
    L.fp_mesh_free.argtypes = [C.c_void_p]
    L.fp_image_read_png.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.fp_frame_size.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    L.fp_read_rgb_depth_mask.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fp_read_cam_k.argtypes = [C.c_char_p, C.c_void_p]
    L.fp_image_write_png_rgb.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    L.fp_draw_bbox3d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fp_net_create.restype = C.c_void_p
    L.fp_net_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.fp_net_destroy.restype = None
    L.fp_net_destroy.argtypes = [C.c_void_p]
    L.fp_net_blob.restype = C.c_void_p
    L.fp_net_blob.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.fp_net_infer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.fp_net_max_batch.argtypes = [C.c_void_p]
    L.fp_mesh_view.restype = C.POINTER(FpMesh)
    L.fp_mesh_view.argtypes = [C.c_void_p]
    vp, ci, cf, cs = C.c_void_p, C.c_int, C.c_float, C.c_char_p
    sigs = {
        "fp_set_inplane_steps": [vp, ci], "fp_num_hypotheses": [vp],
        "fp_register": [vp, vp, vp, vp, ci, ci, cs, ci, vp],
        "fp_track": [vp, vp, vp, ci, ci, vp, cs, ci, vp],
        "fp_register_ex": [vp, vp, vp, vp, ci, ci, ci, cs, ci, vp],
        "fp_track_ex": [vp, vp, vp, ci, ci, ci, vp, cs, ci, vp],
        "fp_track_submit": [vp, vp, vp, ci, ci, ci, vp, cs, ci],
        "fp_track_wait": [vp, vp],
        "fp_track_multi": [vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, vp],
        "fp_upload_frame": [vp, vp, vp, ci, ci, ci], "fp_get_xyz_map": [vp, vp],
        "fp_get_hyp_poses": [vp, vp, ci, vp, vp], "fp_filter_depth": [vp, vp, vp],
        "fp_render_and_transform": [vp, cs, vp, ci, cf, vp, vp, ci],
        "fp_debug_rasterize": [vp, cs, vp, ci, cf, vp, vp],
        "fp_refiner_infer": [vp, vp, vp, ci, ci, vp, vp], "fp_scorer_infer": [vp, vp, vp, ci, ci, vp],
        "fp_refine_post_process": [vp, cs, vp, vp, vp, ci, vp], "fp_argmax": [vp, vp, ci, vp],
        "fp_register_shard_begin": [vp, vp, vp, vp, ci, ci, ci, cs, ci, ci, ci, vp, vp],
        "fp_register_shard_finish": [vp, vp, vp, ci, vp, vp, vp],
        "fp_register_shard_begin_packed": [vp, vp, vp, vp, ci, ci, ci, cs, ci, ci, ci, vp, ci],
        "fp_register_shard_finish_packed": [vp, vp, ci, vp, vp],
        "fp_download": [vp, vp, vp, C.c_size_t],
        "fp_register_sharded": [vp, vp, vp, vp, vp, ci, ci, ci, cs, ci, vp, vp], "fp_device": [vp],
        "fp_profile_enable": [vp, ci], "fp_profile_reset": [vp], "fp_profile_report": [vp, vp, ci],
        "fp_synchronize": [vp], "fp_mesh_orient_bounds": [vp, vp, vp],
        "fp_set_precision": [vp, ci], "fp_get_precision": [vp], "fp_calibrate_fp8": [vp, vp, vp, vp, ci, ci, ci, cs],
        "fp_calibrate": [vp, vp, vp, vp, ci, ci, ci, cs, ci], "fp_calibrate_begin": [vp, ci], "fp_calibrate_add_frame": [vp, vp, vp, vp, ci, ci, ci, cs],
        "fp_calibrate_finish": [vp], "fp_calibrate_abort": [vp], "fp_calibrate_frames": [vp], "fp_get_calibration_blob": [vp, ci, vp, C.c_size_t], "fp_set_calibration_blob": [vp, vp, C.c_size_t],
        "fp_get_calibration": [vp, vp], "fp_set_calibration": [vp, vp], "fp_set_float_model": [vp, ci], "fp_get_float_model": [vp],
    }
    for name, at in sigs.items():
        f = getattr(L, name)
        f.argtypes = at
        f.restype = C.c_int
    L.fp_calibration_size.argtypes = []
    L.fp_calibration_size.restype = C.c_size_t
    _LIB = L
    return L


TEST_LIB_PATH = os.environ.get("FP_TEST_LIB_PATH") or os.path.join(_HERE, "libfoundationpose_amd_test.so")  # override: A/B of two builds (tools/)
_TEST_LIB = None


def test_lib() -> C.CDLL:
    """The same sources built with -DFP_TEST_HOOKS: the product plus the fpt_* kernel-level hooks (unit tests of single
    kernels, A/B switches, ablations, stress and debug helpers).  Used by tests/ and tools/ only; nothing in the
    product path loads it."""
    global _TEST_LIB
    if _TEST_LIB is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise RuntimeError(f"{TEST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _TEST_LIB = C.CDLL(TEST_LIB_PATH)
        _TEST_LIB.fp_last_error.restype = C.c_char_p
    return _TEST_LIB


def use_test_lib() -> None:
    """tools/ only: make lib() load the test build (a superset of the product) so that fpt_* switches act on the very
    library instance that runs the pipeline.  Must be called before the first lib()."""
    global LIB_PATH
    assert _LIB is None, "use_test_lib() must come before the first lib()"
    LIB_PATH = TEST_LIB_PATH


def last_error() -> str:
    return (lib().fp_last_error() or b"").decode("utf-8", "replace")
