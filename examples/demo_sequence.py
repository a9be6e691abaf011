#!/usr/bin/env python3
"""Python twin of examples/fp_demo.cpp: Register the first frame, Track the rest, write poses.txt + box overlays.

    python examples/demo_sequence.py --data test_data/mustard0 --refiner refiner.fpw --scorer scorer.fpw --out out
    python examples/demo_sequence.py --synthetic 8 --out out      # seeded synthetic sequence + synthetic weights
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import FoundationPose, _lib, dataset as D, load_mesh, weights as W  # noqa: E402
from foundationpose_cpp_amd.synthetic import to_colmajor  # noqa: E402


def run(data, refiner, scorer, out, name="mustard", refine_itr=1, plots=False):
    seq = D.Sequence(data)
    mesh = load_mesh(name, seq.mesh_path())
    model = FoundationPose(mesh, seq.K, refiner, scorer, max_input_image_height=max(seq.H, 1080), max_input_image_width=max(seq.W, 1920))
    os.makedirs(out, exist_ok=True)
    poses = []
    with open(os.path.join(out, "poses.txt"), "w") as log:
        def emit(i, pose, rgb, plot):
            log.write(seq.ids[i] + "".join(" %.9g" % v for v in to_colmajor(pose[None])[0]) + "\n")
            if plot:
                img = D.draw_bbox3d(rgb, seq.K, D.convert_pose_mesh2bbox(pose, mesh), mesh.dimension)
                _lib.lib().fp_image_write_png_rgb(os.path.join(out, seq.ids[i] + "_plot.png").encode(), img.ctypes.data, seq.H, seq.W)
        rgb, depth, mask = seq.frame(0, with_mask=True)
        ok, pose = model.Register(rgb, depth, mask, name, refine_itr)
        if not ok:
            raise SystemExit(model.last_error)
        emit(0, pose, rgb, True)
        poses.append(pose)
        t0 = time.perf_counter()
        for i in range(1, len(seq)):
            rgb, depth = seq.frame(i)
            ok, pose = model.Track(rgb, depth, pose, name, refine_itr)
            if not ok:
                raise SystemExit(model.last_error)
            emit(i, pose, rgb, plots or i + 1 == len(seq))
            poses.append(pose)
        if len(seq) > 1:
            print(f"tracked {len(seq) - 1} frames, {(len(seq) - 1) / (time.perf_counter() - t0):.1f} fps including PNG decode")
    model.close()
    return np.stack(poses)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data")
    ap.add_argument("--refiner")
    ap.add_argument("--scorer")
    ap.add_argument("--out", default="fp_demo_out")
    ap.add_argument("--name", default="mustard")
    ap.add_argument("--refine-itr", type=int, default=1)
    ap.add_argument("--plots", action="store_true")
    ap.add_argument("--synthetic", type=int, metavar="N", help="write and use an N-frame synthetic sequence + synthetic weights")
    a = ap.parse_args()
    if a.synthetic:
        d = tempfile.mkdtemp()
        a.data = os.path.join(d, "synthetic0")
        D.write_synthetic_sequence(a.data, a.synthetic)
        a.refiner, a.scorer = os.path.join(d, "refiner.fpw"), os.path.join(d, "scorer.fpw")
        W.pack_synthetic("refiner", a.refiner)
        W.pack_synthetic("scorer", a.scorer)
        print("synthetic sequence:", a.data, "(synthetic weights: poses are not meaningful, only reproducible)")
    run(a.data, a.refiner, a.scorer, a.out, a.name, a.refine_itr, a.plots)
    print("wrote", os.path.join(a.out, "poses.txt"))


if __name__ == "__main__":
    main()
