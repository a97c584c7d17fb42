import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    import torch
    torch.set_num_threads(max(1, min(torch.get_num_threads(), 64)))       # the CPU oracle inside the tests: see tests/helpers.py
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test (full-depth oracle)")


GOLDEN = os.path.join(REPO, 'tests', 'golden')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
