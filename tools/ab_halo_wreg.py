"""conv_halo_wreg_kernel (weights global -> registers) against conv_halo_kernel (weight ring in LDS): bit-identical outputs on the two
3x3 / 40x40 layer shapes, with and without residual, then the time of both at N hypotheses.
   python tools/ab_halo_wreg.py [N]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import _lib  # noqa: E402

_lib.use_test_lib()
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 252
p = lambda t: t.ctypes.data_as(C.c_void_p)  # noqa: E731
rng = np.random.default_rng(0)


def run(NB, Cin, Cout, res, hook, iters=1):
    L.fpt_set_halo_wreg(hook)
    ms = C.c_float(0)
    out = np.zeros((NB, 40, 40, Cout), np.float32)
    rc = L.fpt_conv(p(x), p(w), p(b), p(res) if res is not None else None, NB, 40, 40, Cin, Cout, 3, 3, 1, 1, 40, 40, 1, 0, p(out), iters, C.byref(ms))
    assert rc == 0, _lib.last_error()
    return out, ms.value


for Cin, Cout, ipn in ((128, 128, 2), (256, 256, 1)):
    NB = 40
    x = rng.standard_normal((NB, 40, 40, Cin), dtype=np.float32)
    w = (rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout, dtype=np.float32)
    for use_res in (False, True):
        res = rng.standard_normal((NB, 40, 40, Cout), dtype=np.float32) if use_res else None
        o0, _ = run(NB, Cin, Cout, res, 0)
        o1, _ = run(NB, Cin, Cout, res, 1)
        print(f"{Cin}->{Cout} res={use_res}: bit-identical {np.array_equal(o0, o1)}, max |diff| {np.abs(o0 - o1).max():.3e}, |out| max {np.abs(o0).max():.2f}", flush=True)
    NB = ipn * N
    x = rng.standard_normal((NB, 40, 40, Cin), dtype=np.float32)
    res = rng.standard_normal((NB, 40, 40, Cout), dtype=np.float32)
    fl = 2.0 * NB * 1600 * Cout * 9 * Cin
    for rr in (None, res):
        for rep in range(2):
            t = [run(NB, Cin, Cout, rr, h, 20)[1] for h in (0, 1)]
            print(f"{Cin}->{Cout} NB={NB} res={rr is not None}: LDS ring {t[0] * 1e3:7.1f} us ({fl / t[0] / 1e9:6.0f} TF/s) | registers {t[1] * 1e3:7.1f} us ({fl / t[1] / 1e9:6.0f} TF/s)", flush=True)
L.fpt_set_halo_wreg(0)
