import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def enet_c3k19():
    """models.ENet(19) built for 3 input channels with the seeded synthetic weights + its C-ABI param dict"""
    from helpers import make_model
    return make_model(19, 3, seed=0)


@pytest.fixture(scope="session")
def enet_c4k6():
    """Freiburg-Forest shaped variant (RGB+NIR, 6 classes): BASELINE config C5"""
    from helpers import make_model
    return make_model(6, 4, seed=1)
