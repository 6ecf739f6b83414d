import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def pkg():
    import drone2d_amd
    return drone2d_amd


@pytest.fixture(scope='session')
def oracle():
    from oracle_lib import OracleBackend
    return OracleBackend()


@pytest.fixture(scope='session')
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import drone2d_amd
    from drone2d_amd import _lib
    return _lib.HipBackend()
