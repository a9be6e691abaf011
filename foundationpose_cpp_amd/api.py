"""Host-side mirror of the reference's operator interface over the C ABI.

`FoundationPose` mirrors detection_6d::Base6DofDetectionModel (reference
detection_6d_foundationpose/include/detection_6d_foundationpose/foundationpose.hpp:16-77): Register / Track with the
same argument meaning and the same error behaviour (False + message instead of bool + glog line), plus the
stage-level operators the reference's orchestrator calls (RenderAndTransform, GetHypPoses, SyncInfer, ...), which
the parity tests drive.  Poses are numpy [4,4] (row-major as numpy prints them); conversion to the ABI's
column-major float[16] happens here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .synthetic import Mesh, from_colmajor, to_colmajor

FP_HOST, FP_DEVICE = 0, 1
FP_PREC_F16, FP_PREC_BF16, FP_PREC_FP8, FP_PREC_INT8 = 0, 1, 2, 3   # include/foundationpose_amd.h
CROP = 160


class FoundationPoseError(RuntimeError):
    pass


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def load_mesh(name: str, mesh_file_path: str) -> Mesh:
    """CreateAssimpMeshLoader(name, path) (mesh_loader.hpp:92-93): OBJ + MTL + PNG through the C ABI.
    Raises FoundationPoseError where the reference throws (empty path, unreadable file, no UVs).
    Extra attributes: .orient_bounds [4,4], .dimension [3] (GetOrientBounds / GetObjectDimension)."""
    L = _lib.lib()
    h = L.fp_mesh_load_obj(name.encode(), mesh_file_path.encode())
    if not h:
        raise FoundationPoseError(_lib.last_error())
    try:
        v = L.fp_mesh_view(h).contents
        nv, nf = v.num_vertices, v.num_faces

        def arr(ptr, ctype, shape, dtype):
            n = int(np.prod(shape))
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype).reshape(shape).copy()

        mesh = Mesh(name, arr(v.vertices, C.c_float, (nv, 3), np.float32), arr(v.normals, C.c_float, (nv, 3), np.float32),
                    arr(v.texcoords, C.c_float, (nv, 2), np.float32), arr(v.faces, C.c_uint32, (nf, 3), np.int32),
                    arr(v.texture, C.c_uint8, (v.tex_height, v.tex_width, 3), np.uint8), diameter=float(v.diameter),
                    center=np.array(list(v.center), np.float32))
        ob = np.zeros(16, np.float32)
        dim = np.zeros(3, np.float32)
        L.fp_mesh_orient_bounds(h, _p(ob), _p(dim))
        mesh.orient_bounds = from_colmajor(ob)
        mesh.dimension = dim
        return mesh
    finally:
        L.fp_mesh_free(h)


class FoundationPose:
    """CreateFoundationPoseModel (foundationpose.hpp:99-105) equivalent."""

    def __init__(self, meshes, K, refiner_weights: str | None = None, scorer_weights: str | None = None,
                 max_input_image_height: int = 1080, max_input_image_width: int = 1920, device: int = -1):
        """device: HIP device the model lives on (-1 = the calling thread's current device); every call switches to it and back."""
        self._L = _lib.lib()
        if isinstance(meshes, Mesh):
            meshes = [meshes]
        self.meshes = {m.name: m for m in meshes}
        self._keep = []
        arr = (_lib.FpMesh * len(meshes))()
        for i, m in enumerate(meshes):
            v = np.ascontiguousarray(m.vertices, np.float32)
            n = np.ascontiguousarray(m.normals, np.float32)
            uv = np.ascontiguousarray(m.texcoords[:, :2], np.float32)
            f = np.ascontiguousarray(m.faces, np.uint32)
            tex = np.ascontiguousarray(m.texture, np.uint8)
            self._keep += [v, n, uv, f, tex]
            arr[i].name = m.name.encode()
            arr[i].num_vertices, arr[i].num_faces = len(v), len(f)
            arr[i].vertices, arr[i].normals, arr[i].texcoords = _p(v), _p(n), _p(uv)
            arr[i].faces, arr[i].texture = _p(f), _p(tex)
            arr[i].tex_height, arr[i].tex_width = tex.shape[0], tex.shape[1]
            arr[i].diameter = float(m.diameter)
            arr[i].center = (C.c_float * 3)(*[float(x) for x in m.center])
        self.K = np.ascontiguousarray(K, np.float32)
        h = self._L.fp_create_on(device, C.cast(arr, C.c_void_p), len(meshes), _p(self.K),
                                 refiner_weights.encode() if refiner_weights else None,
                                 scorer_weights.encode() if scorer_weights else None,
                                 max_input_image_height, max_input_image_width)
        if not h:
            raise FoundationPoseError(_lib.last_error())
        self._h = C.c_void_p(h)
        self.last_error = ""

    def close(self):
        if getattr(self, "_h", None):
            self._L.fp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc) -> bool:
        self.last_error = "" if rc == 0 else _lib.last_error()
        return rc == 0

    def _must(self, rc):
        if not self._ok(rc):
            raise FoundationPoseError(self.last_error)

    # ---- Base6DofDetectionModel -----------------------------------------------------------------
    def Register(self, rgb, depth, mask, target_name: str, refine_itr: int = 1):
        """-> (ok, pose[4,4]).  foundationpose.hpp:36-41."""
        rgb, depth, mask = self._frame(rgb, depth, mask)
        if rgb is None:
            return False, None
        out = np.zeros(16, np.float32)
        ok = self._ok(self._L.fp_register(self._h, _p(rgb), _p(depth), _p(mask), depth.shape[0], depth.shape[1],
                                          target_name.encode(), refine_itr, _p(out)))
        return ok, (from_colmajor(out) if ok else None)

    def register_detailed(self, rgb, depth, mask, target_name: str, refine_itr: int = 1):
        """Register through the two ABI halves (`fp_register_shard_begin` over the whole grid + `fp_register_shard_finish`):
        the same kernels as `Register`, but also hands back what the reference keeps internal (foundationpose.cpp:206-228):
        -> (ok, pose[4,4], winner index, scores[N], refined poses[N,4,4], pooled score features[N,512])."""
        rgb, depth, mask = self._frame(rgb, depth, mask)
        if rgb is None:
            return False, None, -1, None, None, None
        n = self.num_hypotheses
        feat, poses = C.c_void_p(), C.c_void_p()
        if not self._ok(self._L.fp_register_shard_begin(self._h, _p(rgb), _p(depth), _p(mask), FP_HOST, depth.shape[0],
                                                        depth.shape[1], target_name.encode(), refine_itr, 0, n,
                                                        C.byref(feat), C.byref(poses))):
            return False, None, -1, None, None, None
        out = np.zeros(16, np.float32)
        idx = C.c_int(-1)
        scores = np.zeros(n, np.float32)
        if not self._ok(self._L.fp_register_shard_finish(self._h, feat, poses, n, _p(out), C.byref(idx), _p(scores))):
            return False, None, -1, None, None, None
        refined = np.zeros((n, 16), np.float32)
        feats = np.zeros((n, 512), np.float32)
        # (fp_download: a copy ordered on the model's own stream -- loading a second HIP runtime through ctypes would not even
        # share stream handles with the library's, and the legacy stream is off limits while any thread captures a hipGraph)
        self._must(self._L.fp_download(self._h, _p(refined), poses, refined.nbytes))
        self._must(self._L.fp_download(self._h, _p(feats), feat, feats.nbytes))
        return True, from_colmajor(out), idx.value, scores, from_colmajor(refined), feats

    def Track(self, rgb, depth, hyp_pose, target_name: str, refine_itr: int = 1):
        """-> (ok, pose[4,4]).  foundationpose.hpp:59-64."""
        rgb, depth, _ = self._frame(rgb, depth, None)
        if rgb is None:
            return False, None
        hyp = to_colmajor(np.asarray(hyp_pose, np.float32))
        out = np.zeros(16, np.float32)
        ok = self._ok(self._L.fp_track(self._h, _p(rgb), _p(depth), depth.shape[0], depth.shape[1], _p(hyp),
                                       target_name.encode(), refine_itr, _p(out)))
        return ok, (from_colmajor(out) if ok else None)

    def track_submit(self, rgb, depth, hyp_pose, target_name: str, refine_itr: int = 1) -> bool:
        """Enqueue a Track (frame upload + refinement) on this model's stream and return; `track_wait` fetches the pose.
        One host thread can keep several models (objects) in flight this way.  The frame arrays are kept alive until the wait."""
        rgb, depth, _ = self._frame(rgb, depth, None)
        if rgb is None:
            return False
        hyp = to_colmajor(np.asarray(hyp_pose, np.float32))
        self._pending_frame = (rgb, depth)
        return self._ok(self._L.fp_track_submit(self._h, _p(rgb), _p(depth), FP_HOST, depth.shape[0], depth.shape[1], _p(hyp),
                                                target_name.encode(), refine_itr))

    def track_wait(self):
        """-> (ok, pose[4,4]) of the last `track_submit`."""
        out = np.zeros(16, np.float32)
        ok = self._ok(self._L.fp_track_wait(self._h, _p(out)))
        self._pending_frame = None
        return ok, (from_colmajor(out) if ok else None)

    def track_multi(self, rgb, depth, hyp_poses, target_names, refine_itr: int = 1):
        """Track of K objects of one frame as one batch -> (ok, poses[K,4,4])."""
        rgb, depth, _ = self._frame(rgb, depth, None)
        if rgb is None:
            return False, None
        hyp = to_colmajor(np.asarray(hyp_poses, np.float32))
        K = len(hyp)
        names = (C.c_char_p * K)(*[n.encode() for n in target_names])
        out = np.zeros((K, 16), np.float32)
        ok = self._ok(self._L.fp_track_multi(self._h, _p(rgb), _p(depth), FP_HOST, depth.shape[0], depth.shape[1], K, _p(hyp),
                                             C.cast(names, C.c_void_p), refine_itr, _p(out)))
        return ok, (from_colmajor(out) if ok else None)

    def _frame(self, rgb, depth, mask):
        # CheckInputArguments (foundationpose.cpp:155-179): sizes must agree
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        if rgb.shape[:2] != depth.shape or (mask is not None and mask.shape != depth.shape):
            self.last_error = "[FoundationPose] Got rgb/depth/mask with different size!"
            return None, None, None
        return rgb, depth, mask

    # ---- stage-level operators -------------------------------------------------------------------
    def set_inplane_steps(self, steps: int):
        self._must(self._L.fp_set_inplane_steps(self._h, steps))

    @property
    def num_hypotheses(self) -> int:
        return self._L.fp_num_hypotheses(self._h)

    def upload_frame(self, rgb, depth):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        self._frame_keep = (rgb, depth)
        self._must(self._L.fp_upload_frame(self._h, _p(rgb), _p(depth), FP_HOST, depth.shape[0], depth.shape[1]))
        self._must(self._L.fp_synchronize(self._h))
        self._hw = depth.shape

    def xyz_map(self):
        out = np.zeros(self._hw + (3,), np.float32)
        self._must(self._L.fp_get_xyz_map(self._h, _p(out)))
        return out

    def filter_depth(self):
        e = np.zeros(self._hw, np.float32)
        b = np.zeros(self._hw, np.float32)
        self._must(self._L.fp_filter_depth(self._h, _p(e), _p(b)))
        return e, b

    def get_hyp_poses(self, mask):
        """FoundationPoseSampler::GetHypPoses -> [N,4,4] or None (False in the reference) on failure."""
        mask = np.ascontiguousarray(mask, np.uint8)
        out = np.zeros((self.num_hypotheses, 16), np.float32)
        n = C.c_int(0)
        if not self._ok(self._L.fp_get_hyp_poses(self._h, _p(mask), FP_HOST, _p(out), C.byref(n))):
            return None
        return from_colmajor(out[:n.value])

    def render_and_transform(self, target_name, poses, crop_ratio):
        """FoundationPoseRenderer::RenderAndTransform -> (render_input, transf_input), each [N,160,160,6] f32."""
        p = to_colmajor(np.asarray(poses, np.float32))
        N = len(p)
        a = np.zeros((N, CROP, CROP, 6), np.float32)
        b = np.zeros((N, CROP, CROP, 6), np.float32)
        self._must(self._L.fp_render_and_transform(self._h, target_name.encode(), _p(p), N, crop_ratio, _p(a), _p(b),
                                                   FP_HOST))
        return a, b

    def debug_rasterize(self, target_name, poses, crop_ratio):
        p = to_colmajor(np.asarray(poses, np.float32))
        N = len(p)
        tri = np.zeros((N, CROP, CROP), np.int32)
        rast = np.zeros((N, CROP, CROP, 4), np.float32)
        self._must(self._L.fp_debug_rasterize(self._h, target_name.encode(), _p(p), N, crop_ratio, _p(tri), _p(rast)))
        return tri, rast

    def refiner_infer(self, render_input, transf_input):
        a = np.ascontiguousarray(render_input, np.float32)
        b = np.ascontiguousarray(transf_input, np.float32)
        N = len(a)
        t = np.zeros((N, 3), np.float32)
        r = np.zeros((N, 3), np.float32)
        self._must(self._L.fp_refiner_infer(self._h, _p(a), _p(b), FP_HOST, N, _p(t), _p(r)))
        return t, r

    def scorer_infer(self, render_input, transf_input):
        a = np.ascontiguousarray(render_input, np.float32)
        b = np.ascontiguousarray(transf_input, np.float32)
        N = len(a)
        s = np.zeros(N, np.float32)
        self._must(self._L.fp_scorer_infer(self._h, _p(a), _p(b), FP_HOST, N, _p(s)))
        return s

    def refine_post_process(self, target_name, poses, trans, rot):
        p = to_colmajor(np.asarray(poses, np.float32))
        t = np.ascontiguousarray(trans, np.float32)
        r = np.ascontiguousarray(rot, np.float32)
        out = np.zeros_like(p)
        self._must(self._L.fp_refine_post_process(self._h, target_name.encode(), _p(p), _p(t), _p(r), len(p), _p(out)))
        return from_colmajor(out)

    def argmax(self, scores) -> int:
        s = np.ascontiguousarray(scores, np.float32).ravel()
        idx = C.c_int(-1)
        self._must(self._L.fp_argmax(self._h, _p(s), len(s), C.byref(idx)))
        return idx.value

    # ---- float model of the rendering stage (1 = contracted like nvcc -fmad=true, default; 0 = separate roundings) ----
    def set_float_model(self, fmad: int):
        self._must(self._L.fp_set_float_model(self._h, int(fmad)))

    @property
    def device(self) -> int:
        return self._L.fp_device(self._h)

    @property
    def float_model(self) -> int:
        return self._L.fp_get_float_model(self._h)

    # ---- network precision (include/foundationpose_amd.h "network precision") ---------------------
    def set_precision(self, precision: int):
        """FP_PREC_F16 (the reference's TensorRT --fp16 engines, default) / FP_PREC_BF16 / FP_PREC_FP8, FP_PREC_INT8 (after calibrate)."""
        self._must(self._L.fp_set_precision(self._h, precision))

    @property
    def precision(self) -> int:
        return self._L.fp_get_precision(self._h)

    def calibrate(self, rgb, depth, mask, target_name: str, precision: int = FP_PREC_INT8):
        """Post-training quantisation of the 8-bit precision on one frame: f16 statistics -> per-channel scales -> bias correction."""
        rgb, depth, mask = self._frame(rgb, depth, mask)
        if rgb is None:
            raise FoundationPoseError(self.last_error)
        self._must(self._L.fp_calibrate(self._h, _p(rgb), _p(depth), _p(mask), FP_HOST, depth.shape[0], depth.shape[1],
                                        target_name.encode(), precision))

    def calibrate_frames(self, frames, target_name: str, precision: int = FP_PREC_INT8):
        """Post-training quantisation over several frames (fp_calibrate_begin / _add_frame / _finish): `frames` = iterable of
        (rgb, depth, mask) or of objects with .rgb / .depth / .mask.  K >= 8 frames of the deployment's scene family make the
        common-mode part of the correction carry over to frames the calibration never saw."""
        self._must(self._L.fp_calibrate_begin(self._h, precision))
        try:
            for f in frames:
                rgb, depth, mask = (f.rgb, f.depth, f.mask) if hasattr(f, "rgb") else f
                rgb, depth, mask = self._frame(rgb, depth, mask)
                if rgb is None:
                    raise FoundationPoseError(self.last_error)
                self._must(self._L.fp_calibrate_add_frame(self._h, _p(rgb), _p(depth), _p(mask), FP_HOST, depth.shape[0], depth.shape[1],
                                                          target_name.encode()))
        except Exception:
            self._L.fp_calibrate_abort(self._h)
            raise
        self._must(self._L.fp_calibrate_finish(self._h))

    def calibrate_fp8(self, rgb, depth, mask, target_name: str):
        self.calibrate(rgb, depth, mask, target_name, FP_PREC_FP8)

    def get_calibration_blob(self, precision: int) -> bytes:
        n = self._L.fp_calibration_size()
        buf = np.zeros(n, np.uint8)
        self._must(self._L.fp_get_calibration_blob(self._h, precision, _p(buf), n))
        return buf.tobytes()

    def set_calibration_blob(self, blob: bytes):
        buf = np.frombuffer(blob, np.uint8).copy()
        self._must(self._L.fp_set_calibration_blob(self._h, _p(buf), buf.size))

    def get_calibration(self):
        out = np.zeros(32, np.float32)
        self._must(self._L.fp_get_calibration(self._h, _p(out)))
        return out

    def set_calibration(self, amax):
        a = np.ascontiguousarray(amax, np.float32)
        assert a.size == 32
        self._must(self._L.fp_set_calibration(self._h, _p(a)))

    # ---- measurement -----------------------------------------------------------------------------
    def profile(self, on: bool):
        self._must(self._L.fp_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._must(self._L.fp_profile_reset(self._h))

    def profile_report(self) -> dict:
        buf = C.create_string_buffer(1 << 16)
        self._must(self._L.fp_profile_report(self._h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms, flops, nbytes = line.split()
            out[name] = dict(calls=int(calls), ms=float(ms), flops=float(flops), bytes=float(nbytes))
        return out

    def synchronize(self):
        self._must(self._L.fp_synchronize(self._h))

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return self._L.fp_stream(self._h)
