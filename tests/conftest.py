import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "compat")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing
    at the first one.  With `-m gpu` (the driver's GPU run) nothing is skipped: a missing device or
    library there must fail loudly, not pass as "skipped"."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    import torch
    so = os.path.join(ROOT, "sparf_amd", "libsparf_hip.so")
    if torch.cuda.is_available() and os.path.exists(so):
        return
    skip = pytest.mark.skip(reason="needs an MI355X and the built libsparf_hip.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    d = os.path.join(ROOT, "tests", "golden")

    def load(name):
        return dict(np.load(os.path.join(d, name + ".npz")))

    return load
