import os
import sys

import pytest

os.environ.setdefault("FD_STRICT", "1")     # training forward / backward: a layer the HIP kernels do not cover raises instead of running on stock ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:     # shared helpers (effnet_init, test_model_gpu.randomize_norms)
    sys.path.insert(1, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))

    return load
