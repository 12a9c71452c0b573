import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # before HIP initialises: the eight-frame pipeline of free-running Redraw()s (cadrays_amd/__init__.py)

import pytest
import torch  # noqa: F401 -- must be imported BEFORE libcadrays_hip.so initialises HIP: torch bundles its own HIP runtime and
              # finds no GPU when it is loaded into a process where the system runtime is already up (seen on the GPU box)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  On a GPU box a missing library is a failure, not a skip."""
    from cadrays_amd._lib import load_library
    return load_library()
