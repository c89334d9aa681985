"""Boundary behaviour of the drop-in module that the reference's callers rely on, on CPU through the numpy op mirror
(tests/fake_backend.py): gradients of an eval-mode forward (frozen-BatchNorm fine-tuning), nn.DataParallel replicas
(reference trainval.py:56-61), BatchNorm momentum / eps read from the modules, and the autograd contract of the fused node."""
import copy
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import load_golden
from fake_backend import use_oracle_ops
from test_plan_cpu import build

NAME = 'j17_a333_c16_dil'


def _model(name=NAME):
    cfg, z, state, grads, post = load_golden(name)
    m = build(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    use_oracle_ops(m)
    return cfg, z, state, m


def test_eval_mode_gradients_match_oracle():
    """backward after an eval-mode forward used uninitialised batch statistics.  Eval BatchNorm is a fixed affine map;
    its gradients must equal the oracle's (reference F.batch_norm(training=False) under autograd)."""
    from oracle import gast_oracle as go
    cfg, z, state, m = _model()
    # running statistics that differ from the batch statistics, so that a training-mode backward would be visibly wrong
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for k, b in m.named_buffers():
            if k.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.2)
            elif k.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
    st = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    om = go.OracleModel(go.adj_from_parents(cfg['parents']), cfg['arc'], cfg['channels'], causal=cfg['causal'], variant=cfg['variant'])
    loss_ref, y_ref, g_ref, _ = om.loss_and_grads(st, z['x'], z['y3d'], training=False)
    m.eval()
    y = m(torch.from_numpy(z['x']))
    loss = torch.mean(torch.norm(y - torch.from_numpy(z['y3d']), dim=-1))
    loss.backward()
    np.testing.assert_allclose(y.detach().numpy(), y_ref, atol=2e-5, rtol=0)
    assert abs(loss.item() - loss_ref) < 1e-5
    for k, p in m.named_parameters():
        r = g_ref[k]
        tol = 2e-4 * max(1e-3, float(np.abs(r).max())) + 2e-5
        assert p.grad is not None and float(np.abs(p.grad.numpy() - r).max()) <= tol, k
    # eval mode must not touch the running statistics
    for k, b in m.named_buffers():
        np.testing.assert_array_equal(b.numpy(), st[k], err_msg=k)


def _replicate_on_cpu(model):
    """What torch.nn.parallel.replicate does to a module tree, without the CUDA broadcast: `_replicate_for_data_parallel()` on every
    module, children rewired, every parameter replaced by a non-leaf copy held as a plain attribute and recorded in
    `_former_parameters` (torch/nn/parallel/replicate.py)."""
    mods = list(model.modules())
    idx = {m: i for i, m in enumerate(mods)}
    reps = []
    for m in mods:
        r = m._replicate_for_data_parallel()
        r._former_parameters = OrderedDict()
        reps.append(r)
    for i, m in enumerate(mods):
        for key, child in m._modules.items():
            if child is None:
                reps[i]._modules[key] = None
            else:
                setattr(reps[i], key, reps[idx[child]])
        for key, p in m._parameters.items():
            if p is None:
                reps[i]._parameters[key] = None
            else:
                c = p * 1.0                      # non-leaf copy: gradients flow back to the source parameter (Broadcast.backward)
                setattr(reps[i], key, c)
                reps[i]._former_parameters[key] = c
        for key, b in m._buffers.items():
            setattr(reps[i], key, b)             # (device 0's replica shares the source buffers)
    return reps[0]


def test_data_parallel_replica_forward_backward():
    """A DataParallel replica has no `_parameters` and shares `_runner` with the other replicas: the forward has to find the
    broadcast tensors by attribute walk, use a per-device engine, and return gradients to autograd."""
    cfg, z, state, m = _model()
    x, y3d = torch.from_numpy(z['x']), torch.from_numpy(z['y3d'])
    m.train()
    ref = copy.deepcopy(m)
    use_oracle_ops(ref)
    y0 = ref(x)
    torch.mean(torch.norm(y0 - y3d, dim=-1)).backward()
    rep = _replicate_on_cpu(m)
    assert rep._is_replica and len(list(rep.parameters())) == 0
    y1 = rep(x)
    torch.mean(torch.norm(y1 - y3d, dim=-1)).backward()
    np.testing.assert_allclose(y1.detach().numpy(), y0.detach().numpy(), atol=1e-6, rtol=0)
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, k
        np.testing.assert_allclose(p.grad.numpy(), q.grad.numpy(), atol=1e-6 + 1e-5 * float(q.grad.abs().max()), rtol=0, err_msg=k)
    # the replica ran on its own engine; the master's engine and packer were not touched
    assert m._runner._engine is None or m._runner._engine is not m._runner.engine_for(x.device)
    # a second forward of the same replica object with fresh copies (what DataParallel does every iteration)
    rep2 = _replicate_on_cpu(m)
    y2 = rep2(x)
    np.testing.assert_allclose(y2.detach().numpy(), y0.detach().numpy(), atol=1e-6, rtol=0)


def test_marked_replica_without_copies_raises_clearly():
    cfg, z, state, m = _model()
    r = m._replicate_for_data_parallel()
    with pytest.raises(RuntimeError, match='_former_parameters'):
        r(torch.from_numpy(z['x']))


def test_batchnorm_momentum_and_eps_are_read_from_the_modules():
    """momentum / eps were module constants of the engine; a `bn.momentum = m` decay schedule had no effect."""
    cfg, z, state, m = _model()
    x = torch.from_numpy(z['x'])
    init = {k: b.clone() for k, b in m.named_buffers()}
    m.train()
    with torch.no_grad():
        m(x)
    d1 = {k: b - init[k] for k, b in m.named_buffers() if k.endswith(('running_mean', 'running_var'))}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 0.3
    with torch.no_grad():
        m(x)
    for k, b in m.named_buffers():
        if k in d1:      # running += momentum * (batch - running): three times the step of momentum 0.1
            np.testing.assert_allclose((b - init[k]).numpy(), 3.0 * d1[k].numpy(), rtol=2e-4, atol=1e-6, err_msg=k)
    # eps: changes the normalisation of the first BatchNorm visibly when it is large
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    with torch.no_grad():
        y_a = m(x)
        m.init_bn.eps = 0.5
        y_b = m(x)
    assert float((y_a - y_b).abs().max()) > 1e-4
    m.init_bn.momentum = None
    with pytest.raises(NotImplementedError):
        m(x)


def test_autograd_contract_errors_are_explicit():
    cfg, z, state, m = _model()
    m.train()
    x = torch.from_numpy(z['x']).requires_grad_(True)
    with pytest.raises(RuntimeError, match='input batch'):
        m(x)
    y = m(torch.from_numpy(z['x']))
    loss = y.sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='second time'):
        loss.backward()


def test_dropout_params_edge_cases():
    from gast_hip.binding import dropout_params
    assert dropout_params(0.0) == (0, 1.0)
    assert dropout_params(1.0) == (65536, 0.0)
    t, k = dropout_params(0.25)
    assert t == 16384 and abs(k - 4.0 / 3.0) < 1e-12
    with pytest.raises(ValueError):
        dropout_params(1.5)


def test_graph_mode_leaves_the_cpu_mirror_path_alone(monkeypatch):
    """The module's hipGraph replay (on by default since round 3, GAST_HIP_GRAPH=0 turns it off) is decided by the runner at
    construction and only ever engages for CUDA inputs on the HIP op set: with the numpy op mirror (or any call that cannot be
    captured) the module keeps its eager path and gives the same result."""
    from tests_helpers import PARENTS
    cfg = dict(J=17, parents=PARENTS[17], arc=[3, 3], channels=16, causal=False, variant='dilated')
    monkeypatch.setenv('GAST_HIP_DTYPE', 'fp32')
    monkeypatch.setenv('GAST_HIP_GRAPH', '0')
    torch.manual_seed(0)
    m0 = build(cfg)
    monkeypatch.delenv('GAST_HIP_GRAPH')
    torch.manual_seed(0)
    m1 = build(cfg)
    assert m1._runner.graph_mode and not m0._runner.graph_mode
    use_oracle_ops(m0)
    use_oracle_ops(m1)
    x = torch.rand(3, 11, 17, 2) * 2 - 1
    outs = []
    for m in (m0, m1):
        m.train()
        ys = [m(x) for _ in range(4)]          # (the fourth call would be a replay on the GPU)
        ys[-1].sum().backward()
        outs.append((ys[-1].detach(), [p.grad.clone() for p in m.parameters()]))
    assert not m1._runner._graphs
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
