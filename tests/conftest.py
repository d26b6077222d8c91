import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture(params=['bf16x3', 'f32'])
def precision(request):
    """Run under both matrix-arithmetic modes of the library (split-bf16 default, exact f32)."""
    from slotformer_amd import _lib
    lib = _lib.lib()
    old = lib.sf_get_precision()
    lib.sf_set_precision(1 if request.param == 'bf16x3' else 0)
    yield request.param
    lib.sf_set_precision(old)
