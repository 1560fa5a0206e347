import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU runs: initialise torch's HIP runtime BEFORE libpoulpy_hip.so is loaded.  The at-scale tests use torch tensors as plumbing;
    torch ships its own libamdhip64 and reports "No HIP GPUs are available" when another copy of the runtime initialised first."""
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ref
    ref.build()
    return ref


def have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
