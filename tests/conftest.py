import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def dsp():
    """the product package (ctypes binding of libdspmap_hip.so); builds the .so if it is missing"""
    sys.path.insert(0, os.path.join(ROOT, "dsp-map_amd"))
    import build_ext
    build_ext.build()
    import dsp_map_amd
    return dsp_map_amd


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py
