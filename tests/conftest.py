import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


@pytest.fixture
def sv_opts():
    """Setter for the library's plan-time options (``sv_set_option``); everything touched goes back to -1 (the
    planner's own choice) after the test."""
    from spectrum_vae_amd import ops
    touched = set()

    def setter(name, value):
        touched.add(name)
        ops.set_option(name, value)

    yield setter
    for name in touched:
        ops.set_option(name, -1)
