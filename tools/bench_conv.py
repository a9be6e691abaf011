#!/usr/bin/env python3
"""Micro-benchmark of conv_igemm_kernel on the layer shapes of one Register (per-GPU batch N hypotheses).
Random fp16 data (guide rule 25: never zero-filled).  Uses the fpt_conv hook of the library (tests-only symbol)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import _lib  # noqa: E402

SHAPES = [  # name, images-per-hyp, H, W, Cin, Cout, k, stride, pad, launches per Register
    ("stem(s2d)", 2, 80, 80, 32, 64, 4, 1, 2, 2),
    ("a1 3x3s2 64->128", 2, 80, 80, 64, 128, 3, 2, 1, 2),
    ("128 3x3", 2, 40, 40, 128, 128, 3, 1, 1, 8),
    ("256 3x3", 1, 40, 40, 256, 256, 3, 1, 1, 8),
    ("b2 3x3s2 256->512", 1, 40, 40, 256, 512, 3, 2, 1, 2),
    ("512 3x3", 1, 20, 20, 512, 512, 3, 1, 1, 8),
    ("qkv 512->1536", 400, 1, 1, 512, 1536, 1, 1, 0, 3),
    ("lin 512->512", 400, 1, 1, 512, 512, 1, 1, 0, 6),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hyps", type=int, default=126)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ablate", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--clock", action="store_true")
    ap.add_argument("--hook", nargs=2, action="append", default=[], metavar=("NAME", "VALUE"), help="test hook to set first, e.g. --hook fpt_set_halo2 0")
    ap.add_argument("--fp8", action="store_true", help="e4m3 operands and output (scales 1/16)")
    ap.add_argument("--res", action="store_true", help="with a residual input (the second conv of a residual block)")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 = 128-px 2-stage kernel, 2 = 256-px 3-stage kernel")
    a = ap.parse_args()
    _lib.use_test_lib()
    L = _lib.lib()
    L.fpt_set_conv_variant(a.variant)
    L.fpt_set_conv_ablate(a.ablate)
    for name, value in a.hook:
        getattr(L, name)(int(value))
    rng = np.random.default_rng(0)
    tot_ms = tot_fl = 0.0
    shapes = [sh for sh in SHAPES if not a.only or a.only in sh[0]]
    for name, ipn, H, W, Cin, Cout, k, stride, pad, launches in shapes:
        NB = ipn * a.hyps
        x = rng.standard_normal((NB, H, W, Cin), dtype=np.float32)
        w = (rng.standard_normal((Cout, k, k, Cin), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
        b = rng.standard_normal(Cout, dtype=np.float32)
        OH = (H + 2 * pad - k) // stride + 1
        OW = (W + 2 * pad - k) // stride + 1
        if k == 4:
            OH, OW = H, W
        out = np.zeros((NB, OH, OW, Cout), np.float32)
        ms = C.c_float(0)
        p = lambda t: t.ctypes.data_as(C.c_void_p)  # noqa: E731
        res = rng.standard_normal(out.shape, dtype=np.float32) if a.res and Cin == Cout and stride == 1 else None
        if a.fp8:
            L.fpt_conv_dt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 13 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
            rc = L.fpt_conv_dt(p(x), p(w), p(b), p(res) if res is not None else None, NB, H, W, Cin, Cout, k, k, stride, pad, OH, OW, 1, 0, p(out), a.iters,
                               C.byref(ms), 2, 2, 1 / 16, 1 / 16, 1 / 16)
            assert rc == 0, _lib.last_error()
            fl = 2.0 * NB * OH * OW * Cout * k * k * Cin
            print(f"{name:22s} [fp8] M={NB * OH * OW:8d} K={k * k * Cin:5d} N={Cout:5d}  {ms.value * 1e3:9.1f} us  {fl / ms.value / 1e9:8.1f} TF/s")
            continue
        rc = L.fpt_conv(p(x), p(w), p(b), p(res) if res is not None else None, NB, H, W, Cin, Cout, k, k, stride, pad, OH, OW, 1, 0, p(out), a.iters,
                        C.byref(ms))
        assert rc == 0, _lib.last_error()
        fl = 2.0 * NB * OH * OW * Cout * k * k * Cin
        if a.clock:
            nblk = 1 << 16
            assert L.fpt_clk_probe(nblk, None, None) == 0
            L.fpt_conv(p(x), p(w), p(b), None, NB, H, W, Cin, Cout, k, k, stride, pad, OH, OW, 1, 0, p(out), 3, C.byref(ms))
            mhz, cyc = C.c_double(0), C.c_double(0)
            L.fpt_clk_probe(-min(nblk, 512), C.byref(mhz), C.byref(cyc))
            print(f"   clock probe: {mhz.value:7.0f} MHz shader clock, {cyc.value:9.0f} cycles per block main loop")
        print(f"{name:22s} M={NB * OH * OW:8d} K={k * k * Cin:5d} N={Cout:5d}  {ms.value * 1e3:9.1f} us  {fl / ms.value / 1e9:8.1f} TF/s")
        tot_ms += ms.value * launches
        tot_fl += fl * launches
    print(f"weighted (per-Register mix): {tot_fl / tot_ms / 1e9:.1f} TF/s, {tot_ms:.3f} ms for N={a.hyps}")


if __name__ == "__main__":
    main()
