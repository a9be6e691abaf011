"""Determinism of one convolution under GPU contention: N host threads, own streams/buffers, bitwise compare (0 = ok)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import _lib
_lib.use_test_lib()
L = _lib.lib()
L.fpt_conv_stress.restype = ctypes.c_longlong
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.fpt_set_conv_variant(variant)
for (NB, H, Cin, Cout, res) in [(126, 40, 128, 128, 1), (126, 40, 256, 256, 1), (126, 40, 256, 256, 0), (252, 20, 512, 512, 1)]:
    for nth, mix in ((1, 0), (2, 0), (2, 1), (4, 1)):
        bad = L.fpt_conv_stress(NB, H, Cin, Cout, res, 40, nth, mix)
        print(f"variant {variant} NB={NB} H={H} {Cin}->{Cout} res={res} threads={nth} mix={mix}: mismatching words = {bad}")
