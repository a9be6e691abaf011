import os
import sys

import pytest

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
