"""Sustained dense fp16 MFMA rate of the whole chip (register-resident v_mfma_f32_16x16x32_f16).

    python tools/mfma_peak.py            # sweeps waves per SIMD, random and all-zero operands
The number bench.py quotes beside the datasheet peak: what the matrix pipes deliver at the clock the power limit allows.
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import _lib
_lib.use_test_lib()
L = _lib.lib()
L.fpt_mfma_peak.restype = C.c_float
L.fpt_mfma_peak.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for zero in (0, 1, 2):   # 2 = random operands, v_mfma_f32_32x32x16_f16 instead of 16x16x32
    for w in (1, 2, 4, 8):
        out = []
        for _ in range(3):
            mhz = C.c_double(0)
            v = L.fpt_mfma_peak(200000, w, zero, C.byref(mhz))
            out.append(f"{v:.0f} TF/s @ {mhz.value:.0f} MHz")
        print(f"{ {0: 'random', 1: 'zero', 2: 'random, 32x32x16'}[zero] } operands, waves/SIMD {w}: " + "   ".join(out))
