import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip silently: only skip when the
    # marker expression did not ask for gpu tests.
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="needs a GPU")
    if not _has_gpu():
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)
