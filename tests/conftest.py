import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the HIP path is the product.
    pass


@pytest.fixture(scope="session")
def lib_built():
    import mgn_amd
    if not os.path.exists(mgn_amd.LIB_PATH):
        from importlib import import_module
        g = import_module("__graft_entry__")
        g.build()
    return mgn_amd.LIB_PATH
