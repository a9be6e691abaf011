"""Build-level guards (no GPU needed):
  * the product and the test library load (every kernel launch stub resolves) and the product exports no fpt_* hook;
  * NO packed-f32 VALU instruction (any v_pk_*_f32, v_pk_mov_b32) in any code object of the product: those returned wrong values in lanes 48-63 whenever waves of another queue's kernel shared the SIMD
    (the root cause of the round-1 'stale read' under two concurrently running models)."""
import ctypes
import os
import re
import shutil
import subprocess
import tempfile

from foundationpose_cpp_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _disassemble(lib_path):
    d = tempfile.mkdtemp()
    try:
        shutil.copy(lib_path, os.path.join(d, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
        text = []
        for f in sorted(os.listdir(d)):
            if "gfx950" in f:
                text.append(subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], cwd=d, check=True, capture_output=True, text=True).stdout)
        return "\n".join(text)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_libraries_load_and_product_has_no_test_hooks():
    L = _lib.lib()
    T = _lib.test_lib()
    assert T.fpt_conv_dt and T.fpt_attention_dt and T.fpt_mfma_peak
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert " fpt_" not in nm, [l for l in nm.splitlines() if "fpt_" in l][:5]
    for sym in ("fpt_set_conv_variant", "fpt_conv"):
        try:
            getattr(L, sym)
            raise AssertionError(f"product exports {sym}")
        except AttributeError:
            pass
    assert isinstance(L, ctypes.CDLL)


def test_no_packed_f32_instructions_in_the_product():
    # no skip: without the disassembler the guard would silently not run (the ROCm image always has it)
    assert os.path.exists(OBJDUMP), f"{OBJDUMP} is missing: the packed-f32 guard cannot run"
    asm = _disassemble(_lib.LIB_PATH)
    kernels = len(re.findall(r"^[0-9a-f]+ <[^>]+>:$", asm, flags=re.M))
    assert kernels >= 60, kernels                      # the extraction really saw the code objects
    assert asm.count("v_mfma_f32_16x16x128_f8f6f4") > 100 and asm.count("v_mfma_f32_16x16x32_bf16") > 100
    bad = re.findall(r"v_pk_\w+_f32|v_pk_mov_b32", asm)
    assert not bad, (len(bad), sorted(set(bad)))
