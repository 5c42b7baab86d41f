import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the oracle library exists (the product .so is built by __graft_entry__.build())."""
    from oracle import oracle as orc

    orc._lib()
    # torch brings its own copy of the HIP runtime: it has to see the GPU before libreadsb_amd.so (linked against
    # /opt/rocm's) initialises it, otherwise torch reports "No HIP GPUs are available" (load order)
    import torch

    if torch.cuda.is_available():
        torch.cuda.init()
    return True
