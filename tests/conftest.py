import os
import sys

import pytest

try:                 # torch wheels bundle their own HIP / HSA runtime: it has to be the FIRST one loaded into the process -- a torch
    import torch     # imported after libfoundationpose_amd.so (which links /opt/rocm's) finds "No HIP GPUs" 
except ImportError:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def syn_mesh():
    from foundationpose_cpp_amd import synthetic as syn
    return syn.make_mesh()


@pytest.fixture(scope="session")
def syn_scene(syn_mesh):
    from foundationpose_cpp_amd import synthetic as syn
    return syn.make_scene(syn_mesh)


DISC_SEED = 9


@pytest.fixture(scope="session")
def disc_cal():
    """calibration record of the DISCRIMINATING synthetic weight set (oracle/disc_weights.py wrote it; numpy only to apply)"""
    from foundationpose_cpp_amd import weights as W
    return W.load_calibration(os.path.join(ROOT, "tests", "golden", f"disc_calib_seed{DISC_SEED}.npz"))


@pytest.fixture(scope="session")
def disc_nets(tmp_path_factory, disc_cal):
    """(refiner.fpw, scorer.fpw, torch refiner, torch scorer) built from ONE state dict each -- outputs differ between
    hypotheses by >= 30 % of their magnitude / score spread >= 0.5 (tests/test_disc_weights_cpu.py holds the oracle to that)"""
    from foundationpose_cpp_amd import weights as W
    from oracle import nets_torch as NT
    d = tmp_path_factory.mktemp("wdisc")
    rp, sp = str(d / "refiner.fpw"), str(d / "scorer.fpw")
    rs = W.pack_synthetic("refiner", rp, DISC_SEED, disc_cal)
    ss = W.pack_synthetic("scorer", sp, DISC_SEED, disc_cal)
    return rp, sp, NT.build("refiner", rs), NT.build("scorer", ss)


def pytest_runtest_setup(item):
    """crash hunts on the GPU box (no gdb there): FP_SEGV_TRACE=<path of tools/_bin/libsegv_trace.so> re-installs a native-backtrace
    SIGSEGV handler before every test (the HIP runtime and faulthandler install their own on the way)."""
    so = os.environ.get("FP_SEGV_TRACE")
    if so:
        import ctypes
        ctypes.CDLL(so).segv_trace_install()
