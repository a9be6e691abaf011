#!/usr/bin/env python3
"""Generates tests/golden/fp_golden_v1.npz -- the committed known-answer vectors for the Register/Track hot path.

The reference holds no golden vectors for this path (its only test prints a pose and checks
nothing), and neither its CUDA sources nor its TensorRT engines can run here.  These vectors are therefore produced by
the CPU oracle (oracle/fp_oracle.c restating the reference file:line by file:line, oracle/nets_torch.py for the two
networks) on the seeded synthetic scene of SURVEY.md §8d.  They pin the oracle against drift and give the HIP path a
fixed target that does not depend on the oracle being importable; when real assets arrive (ONNX weights + the mustard
sequence + a TensorRT pose log) they slot into the same file format.

    python tests/golden/make_golden.py          # rewrites fp_golden_v1.npz and fp_golden_disc_v1.npz (deterministic)
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from foundationpose_cpp_amd import synthetic as syn, weights as W  # noqa: E402
from oracle import fp_oracle as fo  # noqa: E402
from oracle import nets_torch as NT  # noqa: E402

HYP_IDS = np.array([0, 100, 251])       # hypotheses whose full crops are stored
NN_IDS = np.array([0, 37, 100, 251])    # hypotheses pushed through the refiner
N_SMALL = 8                             # BASELINE.json configs[0]: Register with the first 8 grid entries


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    mesh = syn.make_mesh()
    scene = syn.make_scene(mesh)
    om = fo.OracleMesh(mesh)
    K, hw, diam = scene.K, scene.depth.shape, mesh.diameter
    g = {}
    # inputs are regenerated from seeds by the tests; their digests catch generator drift
    g["sha_rgb"], g["sha_depth"], g["sha_mask"] = sha(scene.rgb), sha(scene.depth), sha(scene.mask)
    g["sha_vertices"], g["sha_texture"] = sha(mesh.vertices), sha(mesh.texture)
    g["K"], g["diameter"] = K.astype(np.float32), np.float32(diam)

    grid = fo.rotation_grid(40, 60)
    g["rotation_grid_252"] = grid.astype(np.float32)
    g24 = fo.rotation_grid(40, 15)
    g["rotation_grid_1008_first_last"] = g24[[0, 1, 23, 24, 1007]].astype(np.float32)
    poses = fo.get_hyp_poses(scene.depth, scene.mask, K)
    g["hyp_center"] = poses[0, 12:15].astype(np.float32)
    er, bi = fo.erode_depth(scene.depth), None
    bi = fo.bilateral_filter_depth(er)
    g["depth_eroded_rows"] = er[200:280:8].astype(np.float32)
    g["depth_bilateral_rows"] = bi[200:280:8].astype(np.float32)
    g["xyz_rows"] = fo.depth_to_xyz(scene.depth, K)[200:280:8].astype(np.float32)

    for ratio, tag in ((1.2, "r12"), (1.1, "r11")):
        sel = poses[HYP_IDS]
        g[f"crop_tf_{tag}"] = fo.crop_window_tf(sel, K, ratio, diam).astype(np.float32)
        a, tri, _ = fo.render(om, sel, K, hw, ratio, debug=True)
        b = fo.crop(scene.rgb, scene.depth, K, sel, ratio, diam)
        g[f"tri_id_{tag}"] = tri.astype(np.int32)
        g[f"render_{tag}"] = a.astype(np.float32)
        g[f"transf_{tag}"] = b.astype(np.float32)
    g["hyp_ids"], g["nn_ids"] = HYP_IDS, NN_IDS

    # networks (synthetic weights, seed 7) in PyTorch fp32
    import tempfile
    d = tempfile.mkdtemp()
    rs = W.pack_synthetic("refiner", os.path.join(d, "r.fpw"))   # the state the FPW file holds (BatchNorm folded)
    ss = W.pack_synthetic("scorer", os.path.join(d, "s.fpw"))
    refiner, scorer = NT.build("refiner", rs).eval(), NT.build("scorer", ss).eval()
    with torch.no_grad():
        sel = poses[NN_IDS]
        a = fo.render(om, sel, K, hw, 1.2)
        b = fo.crop(scene.rgb, scene.depth, K, sel, 1.2, diam)
        t, r = refiner(torch.from_numpy(a), torch.from_numpy(b))
        g["refiner_trans"], g["refiner_rot"] = t.numpy().astype(np.float32), r.numpy().astype(np.float32)
        g["refined_poses"] = fo.refine_post_process(sel, t.numpy(), r.numpy(), diam).astype(np.float32)

        # configs[0]: Register over the first N_SMALL hypotheses, refine_itr = 1
        p8 = poses[:N_SMALL]
        a = fo.render(om, p8, K, hw, 1.2)
        b = fo.crop(scene.rgb, scene.depth, K, p8, 1.2, diam)
        t, r = refiner(torch.from_numpy(a), torch.from_numpy(b))
        p8r = fo.refine_post_process(p8, t.numpy(), r.numpy(), diam)
        a = fo.render(om, p8r, K, hw, 1.1)
        b = fo.crop(scene.rgb, scene.depth, K, p8r, 1.1, diam)
        sc = scorer(torch.from_numpy(a), torch.from_numpy(b)).numpy().reshape(-1)
        g["register8_refined"] = p8r.astype(np.float32)
        g["register8_scores"] = sc.astype(np.float32)
        g["register8_best"] = np.int32(fo.argmax(sc))

        # Track from the perturbed ground-truth pose (5 deg / 1 cm, seed 5)
        hyp = syn.to_colmajor(syn.perturb_pose(scene.gt_pose)[None])
        a = fo.render(om, hyp, K, hw, 1.2)
        b = fo.crop(scene.rgb, scene.depth, K, hyp, 1.2, diam)
        t, r = refiner(torch.from_numpy(a), torch.from_numpy(b))
        g["track_in"] = hyp.astype(np.float32)
        g["track_out"] = fo.refine_post_process(hyp, t.numpy(), r.numpy(), diam).astype(np.float32)

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp_golden_v1.npz")
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out) // 1024, "KiB;", len(g), "arrays")

    # second weight set: the DISCRIMINATING one (plain draws of seed 9 + tests/golden/disc_calib_seed9.npz, written by
    # `python -m oracle.disc_weights`), under which outputs differ between hypotheses -- 42 hypotheses (every 6th)
    here = os.path.dirname(os.path.abspath(__file__))
    cal = W.load_calibration(os.path.join(here, "disc_calib_seed9.npz"))
    refiner = NT.build("refiner", W.make_synthetic_state("refiner", 9, cal)).eval()
    scorer = NT.build("scorer", W.make_synthetic_state("scorer", 9, cal)).eval()
    d = {"seed": np.int32(9), "hyp_step": np.int32(6)}
    with torch.no_grad():
        sel = poses[::6]
        a = fo.render(om, sel, K, hw, 1.2)
        b = fo.crop(scene.rgb, scene.depth, K, sel, 1.2, diam)
        t, r = refiner(torch.from_numpy(a), torch.from_numpy(b))
        d["refiner_trans"], d["refiner_rot"] = t.numpy().astype(np.float32), r.numpy().astype(np.float32)
        ref = fo.refine_post_process(sel, t.numpy(), r.numpy(), diam)
        d["refined_poses"] = ref.astype(np.float32)
        a = fo.render(om, ref, K, hw, 1.1)
        b = fo.crop(scene.rgb, scene.depth, K, ref, 1.1, diam)
        sc = scorer(torch.from_numpy(a), torch.from_numpy(b)).numpy().reshape(-1)
        d["scores"], d["best"] = sc.astype(np.float32), np.int32(fo.argmax(sc))
    out = os.path.join(here, "fp_golden_disc_v1.npz")
    np.savez_compressed(out, **d)
    print(out, os.path.getsize(out) // 1024, "KiB;", len(d), "arrays; score std %.3f, top gap %.3f" % (sc.std(), np.sort(sc)[-1] - np.sort(sc)[-2]))


if __name__ == "__main__":
    main()
