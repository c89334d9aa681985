import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'gast-net-3dposeestimation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


def golden_index():
    with open(os.path.join(GOLDEN, 'index.json')) as f:
        return json.load(f)


def golden_names():
    return [k for k in golden_index() if not k.startswith('_')]


def load_golden(name):
    cfg = golden_index()[name]
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    state = {k[len('state/'):]: z[k] for k in z.files if k.startswith('state/')}
    grads = {k[len('grad/'):]: z[k] for k in z.files if k.startswith('grad/')}
    post = {k[len('post/'):]: z[k] for k in z.files if k.startswith('post/')}
    return cfg, z, state, grads, post


@pytest.fixture(scope='session')
def has_gpu():
    import torch
    return torch.cuda.is_available()
