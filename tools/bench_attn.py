#!/usr/bin/env python3
"""A/B of the attention kernels at Register shape: 1 = attention32_kernel, 8 = without its XCD remap, 2 / 3 = the round-1 kernel with / without remap."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import _lib
_lib.use_test_lib()
L = _lib.lib()
L.fpt_attention_bench.restype = ctypes.c_float
L.fpt_attention_bench.argtypes = [ctypes.c_int] * 4
B, T = int(os.environ.get("B", 252)), int(os.environ.get("T", 400))
for v in [int(x) for x in os.environ.get('V', '1,8,2,3').split(',')]:
    ms = L.fpt_attention_bench(B, T, 20, v)
    fl = 4.0 * B * 4 * T * T * 128
    print(f"variant {v}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
