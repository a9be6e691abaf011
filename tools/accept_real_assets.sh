#!/bin/bash
# One-command acceptance run for the day the real assets arrive (the two ONNX files of README.md:72, a dataset in the layout of
# test_data/download.md:6-15, and a pose log of the reference's simple_tests run on the same sequence).  Mirrors
# simple_tests/src/test_foundationpose.cpp:48-104 (Register on the first frame, Track on the rest) and ends in PASS / FAIL:
#
#   tools/accept_real_assets.sh --refiner-onnx refiner_hwc.onnx --scorer-onnx scorer_hwc.onnx --data test_data/mustard0 \
#                               --reference-log reference_run.log [--out out_accept] [--rot-deg 1] [--trans-mm 1] [--refiner-fpw f --scorer-fpw f]
#
#   1. python -m foundationpose_cpp_amd.onnx_reader --check   (structural diff against SURVEY.md Appendix B; stops on a DIFF)
#   2. python -m foundationpose_cpp_amd.weights --onnx         (ONNX initialisers -> FPW1, BatchNorm folded)
#   3. builds examples/fp_demo.cpp against libfoundationpose_amd.so and runs it on the sequence -> <out>/poses.txt
#   4. tools/compare_pose_log.py <out>/poses.txt <reference log> --rot-deg R --trans-mm T       (exit 1 over the gate)
# --refiner-fpw / --scorer-fpw skip steps 1-2 (weights already converted; the synthetic dry run of tests/test_accept_script_gpu.py uses it
# for one leg and ONNX files written by tests/onnx_writer.py for the other).  Exit codes: 0 PASS, 1 FAIL (pose gate), 2 usage / a step broke.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=out_accept; ROT=1; TRANS=1; RONNX=""; SONNX=""; RFPW=""; SFPW=""; DATA=""; REFLOG=""
while [ $# -gt 0 ]; do
  case "$1" in
    --refiner-onnx) RONNX=$2; shift 2;; --scorer-onnx) SONNX=$2; shift 2;;
    --refiner-fpw) RFPW=$2; shift 2;; --scorer-fpw) SFPW=$2; shift 2;;
    --data) DATA=$2; shift 2;; --reference-log) REFLOG=$2; shift 2;; --out) OUT=$2; shift 2;;
    --rot-deg) ROT=$2; shift 2;; --trans-mm) TRANS=$2; shift 2;;
    *) echo "unknown argument $1" >&2; exit 2;;
  esac
done
if [ -z "$DATA" ] || [ -z "$REFLOG" ] || { [ -z "$RONNX" ] && [ -z "$RFPW" ]; } || { [ -z "$SONNX" ] && [ -z "$SFPW" ]; }; then
  sed -n 2,16p "$0" >&2; exit 2
fi
mkdir -p "$OUT" || exit 2
step() { echo "== $*"; }
export PYTHONPATH="$ROOT${PYTHONPATH:+:$PYTHONPATH}"
for kind in refiner scorer; do
  if [ $kind = refiner ]; then onnx=$RONNX; fpw=$RFPW; else onnx=$SONNX; fpw=$SFPW; fi
  if [ -z "$fpw" ]; then
    step "1. structural check of $onnx"
    python -m foundationpose_cpp_amd.onnx_reader --check "$onnx" $kind || { echo "FAIL: $onnx is not the architecture this library implements (see the DIFF lines)"; exit 2; }
    step "2. $onnx -> $OUT/$kind.fpw"
    python -m foundationpose_cpp_amd.weights --onnx $kind "$onnx" "$OUT/$kind.fpw" || { echo "FAIL: weight conversion of $onnx"; exit 2; }
    fpw="$OUT/$kind.fpw"
  fi
  if [ $kind = refiner ]; then RFPW=$fpw; else SFPW=$fpw; fi
done
step "3. fp_demo on $DATA"
LIBDIR="$ROOT/foundationpose_cpp_amd"
[ -f "$LIBDIR/libfoundationpose_amd.so" ] || python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" || exit 2
g++ -std=c++17 -I "$ROOT/include" "$ROOT/examples/fp_demo.cpp" -o "$OUT/fp_demo" -L "$LIBDIR" -lfoundationpose_amd "-Wl,-rpath,$LIBDIR" -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib || exit 2
"$OUT/fp_demo" --data "$DATA" --refiner "$RFPW" --scorer "$SFPW" --out "$OUT" > "$OUT/fp_demo.log" 2>&1 || { tail -5 "$OUT/fp_demo.log"; echo "FAIL: fp_demo"; exit 2; }
step "4. pose log against the reference (gate: $ROT deg / $TRANS mm)"
python "$ROOT/tools/compare_pose_log.py" "$OUT/poses.txt" "$REFLOG" --rot-deg "$ROT" --trans-mm "$TRANS"
rc=$?
if [ $rc -eq 0 ]; then echo "PASS: every frame within $ROT deg / $TRANS mm of the reference"; exit 0; fi
if [ $rc -eq 1 ]; then echo "FAIL: pose gate ($ROT deg / $TRANS mm) exceeded"; exit 1; fi
echo "FAIL: the pose logs could not be compared"; exit 2
