"""bench.py's per-kernel cost models must accept every call the engine makes (a signature drift there broke the default
`python bench.py` once): run one training step of the host plan on the numpy op mirror with the KernelTimer wrapped around it
and dummy events."""
import sys
import os

import torch

from conftest import ROOT
from fake_backend import use_oracle_ops
from test_plan_cpu import build
from tests_helpers import PARENTS


class _DummyEvent:
    def __init__(self, enable_timing=True):
        pass

    def record(self):
        pass

    def elapsed_time(self, other):
        return 1.0


def test_kernel_timer_cost_models_cover_the_engine_calls(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(torch.cuda, 'Event', _DummyEvent)
    monkeypatch.setenv('GAST_HIP_DTYPE', 'fp32')
    cfg = dict(J=17, parents=PARENTS[17], arc=[3, 3], channels=16, causal=False, variant='dilated')
    m = build(cfg, dropout=0.1)
    use_oracle_ops(m)
    ops = m._runner.engine.ops
    ops.adam_step = lambda *a, **k: None          # op set entries the numpy mirror has no use for
    ops.run_pack = getattr(ops, 'run_pack', lambda *a, **k: None)
    ops.run_unpack = getattr(ops, 'run_unpack', lambda *a, **k: None)
    for name in ('colsum', 'bn_finalize', 'bn_bwd_finalize', 'wgrad'):
        assert hasattr(ops, name), name
    timer = bench.KernelTimer(ops)
    timer.enabled = True
    m.train()
    x = torch.rand(4, 11, 17, 2) * 2 - 1
    y = m(x)
    y.sum().backward()
    agg = timer.summary()
    for name in ('gemm', 'gemm_multi', 'wgrad_multi', 'semch_agg_fwd', 'semch_agg_bwd', 'attn_fwd', 'attn_bwd', 'bnrelu_apply',
                 'bn_finalize_multi', 'expand_bwd'):
        assert name in agg and agg[name]['launches'] > 0, name
    assert agg['gemm']['bytes'] > 0 and agg['gemm_multi']['flops'] > 0 and agg['wgrad_multi']['bytes'] > 0


def test_stock_operator_baseline_leg_runs_on_cpu():
    """bench.py's cpu_baseline / --stock-baseline legs (the oracle restatement on stock PyTorch operators): one tiny training step
    on CPU, so that a signature drift in oracle/ cannot break the default `python bench.py` on the GPU box."""
    sys.path.insert(0, ROOT)
    import bench
    n, dt = bench._stock_steps('cpu', False, 4, 0.0, 1, warm_B=2)
    assert n == 1 and dt > 0
