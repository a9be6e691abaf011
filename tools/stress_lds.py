"""Does any convolution schedule write outside its own LDS allocation?  A canary kernel co-resides on the CUs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import _lib
_lib.use_test_lib()
L = _lib.lib()
L.fpt_lds_canary.restype = ctypes.c_longlong
for variant in (0, 8, 1):
    L.fpt_set_conv_variant(variant)
    for (NB, H, Cin, Cout) in [(126, 40, 128, 128), (126, 40, 256, 256), (252, 20, 512, 512)]:
        for cb in (2048, 16384):
            bad = L.fpt_lds_canary(NB, H, Cin, Cout, 60, cb)
            print(f"variant {variant} H={H} {Cin}->{Cout} canary {cb} B: corrupted canary words seen = {bad}")
