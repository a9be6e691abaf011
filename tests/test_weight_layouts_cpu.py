"""The weight copies the kernels stream from (DESIGN.md sections 4.5 / 8: MFMA-fragment order for conv_smallx_kernel, LDS-stage order
for gemm_k32 / conv_halo / conv_halo8 / conv_big_pp / conv_deep / conv_pp) against the row-major copy, on the HOST: for every
(row tile, K-step, wave, piece, lane) the address the kernel forms must hold the 16 bytes that lane would have fetched from the
row-major rows (same row, same swizzled chunk).  No GPU needed; the GPU suite then checks the kernels bit for bit."""
import ctypes as C

import pytest

from foundationpose_cpp_amd import _lib


@pytest.mark.parametrize("Cout,row_bytes", [(256, 1024),      # Linear 512 -> 256-row tile
                                            (1536, 1024),     # QKV projection
                                            (256, 4608),      # 3x3, 256 channels
                                            (512, 9216),      # 3x3, 512 channels
                                            (256, 1152)])     # FP8 3x3, 128 channels (1 byte per element)
def test_every_kernel_address_into_the_weight_copies_holds_the_row_major_bytes(Cout, row_bytes):
    T = _lib.test_lib()
    T.fpt_check_weight_layouts.restype = C.c_longlong
    T.fpt_check_weight_layouts.argtypes = [C.c_int, C.c_int]
    assert T.fpt_check_weight_layouts(Cout, row_bytes) == 0


def test_unsupported_sizes_are_refused():
    T = _lib.test_lib()
    T.fpt_check_weight_layouts.restype = C.c_longlong
    T.fpt_check_weight_layouts.argtypes = [C.c_int, C.c_int]
    assert T.fpt_check_weight_layouts(100, 1024) == -1
