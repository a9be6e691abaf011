#!/usr/bin/env python3
"""Per-frame rotation / translation error between two pose logs; non-zero exit when any frame exceeds the gate.

    python tools/compare_pose_log.py ours/poses.txt reference.log --rot-deg 1 --trans-mm 1

This is the acceptance check of the north star ("pose within 1 deg / 1 mm of the TensorRT fp16 reference on mustard") for the day
the real assets are available: run `examples/fp_demo` (writes <out>/poses.txt) and the reference's `simple_tests` on the same
sequence, and hand both logs to this tool.  Either argument may be in either format (auto-detected per file):

  * fp_demo's `poses.txt`: one line per frame, `<frame id> <16 floats, COLUMN-major 4x4>`;
  * the reference's glog output (`simple_tests/src/test_foundationpose.cpp:62,89`: `LOG(WARNING) << "first Pose : " << out_pose;` /
    `"Track pose : "`): Eigen prints the 4x4 ROW by row -- the first row on the line of the label, three more lines after it.
    Frames are numbered in order of appearance (first Pose = frame 0).

Frames are matched by position (frame ids are reported when a log has them).  numpy only.
"""
from __future__ import annotations

import argparse
import re
import sys

import numpy as np

_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|nan|inf)"
_LABEL = re.compile(r"(first Pose|Track pose)\s*:\s*(.*)$")


def _floats(text):
    return [float(t) for t in re.findall(_NUM, text)]


def parse_pose_log(path: str):
    """-> (ids [F], poses [F,4,4] row-major numpy).  Raises ValueError with the offending line on malformed input."""
    with open(path, "r", errors="replace") as f:
        lines = f.read().splitlines()
    ids, poses = [], []
    if any(_LABEL.search(l) for l in lines):                      # the reference's glog / Eigen format
        i = 0
        while i < len(lines):
            m = _LABEL.search(lines[i])
            if not m:
                i += 1
                continue
            rows = [_floats(m.group(2))]
            j = i + 1
            while len(rows) < 4 and j < len(lines):
                if lines[j].strip():
                    rows.append(_floats(lines[j]))
                j += 1
            if len(rows) != 4 or any(len(r) != 4 for r in rows):
                raise ValueError(f"{path}:{i + 1}: expected a 4x4 matrix after '{m.group(1)}', got {rows}")
            ids.append(f"{'register' if m.group(1) == 'first Pose' else 'track'}#{len(ids)}")
            poses.append(np.array(rows, np.float64))
            i = j
    else:                                                          # fp_demo: id + 16 column-major floats
        for n, l in enumerate(lines):
            t = l.split()
            if not t or t[0].startswith("#"):
                continue
            if len(t) != 17:
                raise ValueError(f"{path}:{n + 1}: expected '<id> <16 floats>', got {len(t)} fields")
            ids.append(t[0])
            poses.append(np.array([float(x) for x in t[1:]], np.float64).reshape(4, 4).T)
    if not poses:
        raise ValueError(f"{path}: no poses found")
    return ids, np.stack(poses)


def pose_errors(a: np.ndarray, b: np.ndarray):
    """rotation angle (deg) and translation distance (same unit as the poses) between pose stacks [F,4,4]"""
    d = np.linalg.norm((a[:, :3, :3] - b[:, :3, :3]).reshape(len(a), 9), axis=1)      # |Ra - Rb|_F = 2 sqrt(2) sin(angle / 2)
    ang = np.degrees(2 * np.arcsin(np.clip(d / (2 * np.sqrt(2)), 0, 1)))
    return ang, np.linalg.norm(a[:, :3, 3] - b[:, :3, 3], axis=1)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("log_a")
    ap.add_argument("log_b")
    ap.add_argument("--rot-deg", type=float, default=1.0, help="gate on the rotation error per frame (degrees)")
    ap.add_argument("--trans-mm", type=float, default=1.0, help="gate on the translation error per frame (mm; poses are in metres)")
    ap.add_argument("--quiet", action="store_true", help="summary only")
    a = ap.parse_args(argv)
    try:
        ids_a, pa = parse_pose_log(a.log_a)
        ids_b, pb = parse_pose_log(a.log_b)
    except (OSError, ValueError) as e:
        print(f"error: {e}", file=sys.stderr)
        return 2
    if len(pa) != len(pb):
        print(f"error: {a.log_a} has {len(pa)} poses, {a.log_b} has {len(pb)}", file=sys.stderr)
        return 2
    ang, dist = pose_errors(pa, pb)
    mm = dist * 1e3
    bad = (ang > a.rot_deg) | (mm > a.trans_mm) | ~np.isfinite(ang) | ~np.isfinite(mm)
    if not a.quiet:
        print(f"{'frame':>6} {'id a':>14} {'id b':>14} {'rot [deg]':>10} {'trans [mm]':>11}")
        for i in range(len(pa)):
            print(f"{i:6d} {ids_a[i]:>14} {ids_b[i]:>14} {ang[i]:10.4f} {mm[i]:11.4f}{'   <-- exceeds the gate' if bad[i] else ''}")
    print(f"{len(pa)} frames: rotation max {ang.max():.4f} / mean {ang.mean():.4f} deg, translation max {mm.max():.4f} / mean {mm.mean():.4f} mm; "
          f"gate {a.rot_deg} deg / {a.trans_mm} mm: {'FAIL (%d frames)' % int(bad.sum()) if bad.any() else 'PASS'}")
    return 1 if bad.any() else 0


if __name__ == "__main__":
    sys.exit(main())
