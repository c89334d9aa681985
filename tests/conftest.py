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


# ---- order / state screens for the kernel suites (scripts/stress_kernel_suite.sh drives them on the GPU box) -------------------
#   --gast-shuffle N  (or GAST_TEST_SHUFFLE=N): run the collected tests in the order random.Random(N) shuffles them into -- a
#                     test that passes alone and fails "somewhere in a full run" depends on state an EARLIER launch left behind;
#   GAST_TEST_POISON=1: every torch.empty / empty_like / new_empty the tests and the binding make arrives filled with NaN (float
#                     types) or 0x7f bytes, and the split-K workspace is re-filled with NaN before every GEMM call -- a kernel that
#                     reads an element it (or its finish pass) did not write this launch now fails loudly instead of inheriting a
#                     plausible value from the allocator.
def pytest_addoption(parser):
    parser.addoption('--gast-shuffle', action='store', default=os.environ.get('GAST_TEST_SHUFFLE'),
                     help='shuffle the collected tests with this seed')


def pytest_collection_modifyitems(config, items):
    seed = config.getoption('--gast-shuffle')
    if seed not in (None, ''):
        import random
        random.Random(int(seed)).shuffle(items)


_POISONED = [False]


def poison_allocations():
    """GAST_TEST_POISON=1 (idempotent): patch torch's uninitialised allocators for this process."""
    if _POISONED[0] or os.environ.get('GAST_TEST_POISON', '0') in ('0', ''):
        return _POISONED[0]
    import torch
    _POISONED[0] = True

    def fill(t):
        if t.is_cuda and t.numel():
            if t.dtype.is_floating_point:
                t.fill_(float('nan'))
            elif t.dtype in (torch.int32, torch.int64, torch.int16, torch.uint8, torch.int8):
                t.view(torch.uint8).fill_(0x7f) if t.is_contiguous() else t.fill_(0x7f)
        return t

    for name in ('empty', 'empty_like', 'empty_strided'):
        orig = getattr(torch, name)
        setattr(torch, name, (lambda o: lambda *a, **k: fill(o(*a, **k)))(orig))
    orig_new_empty = torch.Tensor.new_empty
    torch.Tensor.new_empty = lambda self, *a, **k: fill(orig_new_empty(self, *a, **k))
    from gast_hip import binding
    orig_ws = binding.HipOps._splitk_ws

    def ws(self, dev):
        w = orig_ws(self, dev)
        w.fill_(float('nan'))
        return w
    binding.HipOps._splitk_ws = ws
    return True


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
