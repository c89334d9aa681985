"""The differentiable composition of the stand-alone sub-modules (gast_hip/modules.py + gast_hip/autograd_ops.py; SURVEY.md section 8
row f3) on CPU: the four autograd blocks run on the numpy mirror of the op set (tests/fake_backend.py, injected here by monkeypatching
the modules' op-set globals -- the product has no such switch), so the HOST side -- which kernels are paired, operand layouts, the
torch glue that folds / unfolds the parameters -- is pinned to the reference-generated gradients without a GPU.  The kernels themselves
are checked on the GPU box (tests/test_modules_gpu.py, tests/test_kernels_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from test_modules_gpu import CASES, GOLD, build


@pytest.fixture
def mirror(monkeypatch):
    from fake_backend import OracleOps
    import gast_hip.modules as gm
    import gast_hip.autograd_ops as ga
    ops = OracleOps()
    monkeypatch.setattr(gm, '_OPS', ops)
    monkeypatch.setattr(ga, '_OPS', ops)
    monkeypatch.setattr(gm, '_check', lambda mod, x, ndim: x.contiguous().float())       # (the product refuses CPU tensors)
    return ops


@pytest.mark.parametrize('name', CASES)
def test_module_gradients_on_the_numpy_mirror(name, mirror):
    fx = dict(np.load(os.path.join(GOLD, name + '.npz')))
    mod = build(name, fx)
    mod.load_state_dict({k[len('state/'):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('state/')}, strict=True)
    mod.train()
    x = torch.from_numpy(fx['x']).requires_grad_(True)
    y = mod(x)
    assert y.requires_grad
    err = float(np.abs(y.detach().numpy() - fx['y_train']).max())
    assert err <= 1e-4 * max(1.0, float(np.abs(fx['y_train']).max())), (name, err)
    (y * torch.from_numpy(fx['dy'])).sum().backward()
    ref = {k[len('grad/'):]: v for k, v in fx.items() if k.startswith('grad/')}
    gmax = max(float(np.abs(v).max()) for v in ref.values())
    worst = ('', 0.0)
    for k, p in list(mod.named_parameters()) + [('<input>', x)]:
        r = fx['dx'] if k == '<input>' else ref[k]
        assert p.grad is not None, (name, k)
        e = float(np.abs(p.grad.numpy() - r).max()) / (2e-4 * float(np.abs(r).max()) + 2e-5 * gmax)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] <= 1.0, (name, worst)
    sd = mod.state_dict()
    for k, v in fx.items():
        if k.startswith('post/'):
            assert np.allclose(sd[k[len('post/'):]].numpy(), v, rtol=1e-4, atol=1e-5), (name, k)


def test_no_grad_path_is_the_fused_forward_plan(mirror):
    """under torch.no_grad() the fused forward plan runs (no graph recorded), and both paths agree"""
    fx = dict(np.load(os.path.join(GOLD, 'mod_local_j17_c32.npz')))
    mod = build('mod_local_j17_c32', fx)
    mod.load_state_dict({k[len('state/'):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('state/')}, strict=True)
    mod.eval()
    x = torch.from_numpy(fx['x'])
    with torch.no_grad():
        y0 = mod(x)
    y1 = mod(x)
    assert not y0.requires_grad and y1.requires_grad
    assert float((y0 - y1.detach()).abs().max()) < 1e-5
