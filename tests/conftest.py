import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "autograd: the test differentiates through the product (others run under no_grad)")


@pytest.fixture(autouse=True)
def _inference_mode_unless_marked(request):
    """Sampling / validation in the reference run under torch.no_grad() (diffusion_qm9.py:347, Lightning's eval loop); the
    product is differentiable whenever autograd is recording, so value-only tests switch recording off like those callers
    do.  Tests marked `autograd` (tests/test_gpu_training.py) keep it on."""
    import torch
    if request.node.get_closest_marker("autograd"):
        yield
        return
    with torch.no_grad():
        yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
