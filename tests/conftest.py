import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The CPU oracle of the parity tests works on small tensors (a few molecules, hundreds of edge rows): on the GPU boxes' 128-core host
# torch's default intra-op pool makes it 10 x SLOWER than eight threads (test_full_length_chain_production_width: 235 s of a 520 s
# tier on one box, ~20 s with eight).  Eight threads for this process and for the scripts the tests start (tests/fuzz_*.py, bench.py).
os.environ.setdefault("OMP_NUM_THREADS", "8")


def pytest_configure(config):
    try:
        import torch
        torch.set_num_threads(min(8, torch.get_num_threads()))
    except Exception:
        pass
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
