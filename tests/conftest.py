import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the plain-C checker is compiled here, at session start, before any test can have initialised the GPU (compiling is a
    # fork + exec; oracle/c_oracle.py refuses to do that once HIP is up)
    try:
        from oracle import c_oracle
        c_oracle.build()
    except Exception as e:          # no gcc: the tests that need the C port fail with the reason, the others run
        print(f"conftest: C oracle not built: {e!r}")


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
