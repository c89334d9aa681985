"""Why the benchmark's arithmetic is bf16x3 and not plain bf16 -- shown on the REFERENCE's own operator sequence, on CPU.

The restatement of the reference (oracle/gast_oracle.py on oracle/torch_ops.py: F.conv2d / matmul / F.batch_norm ...) is run in
train mode on reference-generated weights with bf16 rounding injected at the matrix-product operands only (storage and everything
else stay fp32 -- the best any bf16-operand implementation, e.g. torch.autocast(bfloat16) of the reference itself, can do):
the outputs move by several 1e-2, above BASELINE.json's 1e-2 bf16 tolerance, whichever of the two operand kinds (activations,
weights) is rounded.  With each operand split into a bf16 hi/lo pair (what GAST_HIP_DTYPE=bf16x3 does inside the MFMA loop) the
same forward stays within 1e-4.  Eval mode looks better only because the untrained net's eval outputs are 100x smaller."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import gast_oracle as go
from oracle import torch_ops as T


def _r(t):
    return t.bfloat16().to(t.dtype)


def _h(t):
    return t.half().to(t.dtype)


def _split(t):
    hi = _r(t)
    return hi + _r(t - hi)


def _backend(act, wt):
    """torch_ops with `act` / `wt` applied to the activation / weight operand of every matrix product (None = leave exact)."""
    m = types.ModuleType('rounded_ops')
    for k in dir(T):
        if not k.startswith('__'):
            setattr(m, k, getattr(T, k))
    ident = lambda t: t     # noqa: E731
    fa, fw = act or ident, wt or ident

    def is_act(t):          # data tensors carry the (B, T, ...) batch axes; parameters and parameter-only adjacencies do not
        return t.ndim >= 4 or (t.ndim == 3 and t.shape[0] > 64)

    def matmul(a, b):
        return T.Var(torch.matmul(fa(a.v) if is_act(a.v) else fw(a.v), fa(b.v) if is_act(b.v) else fw(b.v)))

    def conv2d_k1(x, w, dilation=1, stride=1):
        return T.Var(F.conv2d(fa(x.v), fw(w.v), stride=(stride, 1), dilation=(dilation, 1)))

    def conv1d_1x1(x, w, b):
        return T.Var(F.conv1d(fa(x.v), fw(w.v), b.v))
    m.matmul, m.conv2d_k1, m.conv1d_1x1 = matmul, conv2d_k1, conv1d_1x1
    return m


def _forward(backend, cfg, state, x, training):
    om = go.OracleModel(go.adj_from_parents(cfg['parents']), cfg['arc'], cfg['channels'], causal=cfg['causal'], variant=cfg['variant'],
                        dtype=np.float64)
    with torch.no_grad(), go.use_backend(backend):
        y, _ = om.forward(state, x, training=training)
    return y.v


@pytest.mark.parametrize('name', ['j17_a333_c16_dil', 'j19_a33_c32_dil'])
def test_bf16_operand_rounding_floor_of_the_reference_operators(name):
    cfg, z, state, grads, post = load_golden(name)
    x = z['x']
    exact = _forward(T, cfg, state, x, True)
    assert float((exact - torch.from_numpy(z['y_train'])).abs().max()) < 1e-4       # the float64 restatement IS the reference
    err = {}
    for tag, act, wt in (('bf16 activations+weights', _r, _r), ('bf16 activations only', _r, None), ('bf16 weights only', None, _r),
                         ('split activations, bf16 weights', _split, _r), ('hi/lo split of both (bf16x3)', _split, _split),
                         ('fp16 activations+weights', _h, _h)):
        err[tag] = float((_forward(_backend(act, wt), cfg, state, x, True) - exact).abs().max())
    # any single bf16-rounded operand kind already breaks the 1e-2 bound in train mode ...
    assert err['bf16 activations+weights'] > 1e-2, err
    assert err['bf16 activations only'] > 5e-3 and err['bf16 weights only'] > 5e-3, err
    assert err['split activations, bf16 weights'] > 5e-3, err
    # ... and the hi/lo split of both restores fp32-class agreement
    assert err['hi/lo split of both (bf16x3)'] < 1e-4, err
    # ... while IEEE binary16 operands (11 significand bits instead of 8; VERDICT round 3, Weak #12: 4.2e-3 / 2.2e-3 on these two
    # goldens) stay INSIDE the 16-bit bound at one matrix product per operand pair: the arithmetic of GAST_HIP_DTYPE=f16
    assert err['fp16 activations+weights'] < 1e-2, err
    assert err['fp16 activations+weights'] < err['bf16 activations+weights'] / 4, err
    # eval mode: same relative error, but the untrained net's eval outputs are tiny (running statistics 0 / 1)
    ev = _forward(T, cfg, state, x, False)
    ev16 = _forward(_backend(_r, _r), cfg, state, x, False)
    rel_eval = float((ev16 - ev).abs().max() / ev.abs().max())
    rel_train = err['bf16 activations+weights'] / float(exact.abs().max())
    assert 0.1 < rel_eval / rel_train < 10.0, (rel_eval, rel_train)
