import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    # torch first: its wheel bundles its own HIP runtime, and a process that has already initialised the system's
    # (through librt_amd.so) cannot initialise a second one ("No HIP GPUs are available").  Loaded in this order,
    # librt_amd.so binds to the runtime torch brought, as in bench.py, and device pointers are shared.
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    from raytracer_amd import capi
    c = capi.Context(0)
    yield c
    c.close()
