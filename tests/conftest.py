import os
import sys

import pytest

# exercise the coalesced heat-map tile kernel on the small test fans too (default threshold: 6000 rays)
os.environ.setdefault('RT2D_HEATMAP_FAST_MIN_RAYS', '0')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a host without a GPU skips the gpu-marked tests instead of failing them."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='needs a GPU (MI355X): run with -m gpu on the GPU box')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def rt():
    import rt2d_loader
    return rt2d_loader.load()


@pytest.fixture(scope='session')
def orc():
    from oracle import oracle
    oracle.compile()
    return oracle
