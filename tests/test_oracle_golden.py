"""Pin the numpy oracle against fixtures produced by the reference itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import gast_oracle as go


def _model(cfg, dtype=np.float64):
    adj = go.adj_from_parents(cfg['parents'])
    return go.OracleModel(adj, cfg['arc'], cfg['channels'], causal=cfg['causal'], dropout=0.0,
                          variant=cfg['variant'], dtype=dtype)


def test_adjacency_known_answer():
    import os
    from conftest import GOLDEN
    ref = np.load(os.path.join(GOLDEN, 'adj_j17.npy'))
    ours = go.adj_from_parents([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15])
    np.testing.assert_allclose(ours, ref, rtol=0, atol=1e-7)


def test_patterns_known_answer():
    # SURVEY.md App. B (probe of the reference): J=17 sym nnz 29, con nnz 54; 19 -> 33/62; 15 -> 27/47
    from tests_helpers import PARENTS
    for J, (ns, nc) in {17: (29, 54), 19: (33, 62), 15: (27, 47)}.items():
        s, c = go.local_graph_adjacencies(go.adj_from_parents(PARENTS[J]))
        assert int((s > 0).sum()) == ns and int((c > 0).sum()) == nc
    with pytest.raises(KeyError):
        go.local_graph_adjacencies(np.eye(14))


@pytest.mark.parametrize('name', golden_names())
def test_eval_forward_matches_reference(name):
    cfg, z, state, grads, post = load_golden(name)
    m = _model(cfg)
    assert m.receptive_field() == cfg['receptive_field']
    y, _ = m.forward(state, z['x'], training=False)
    assert y.v.shape == z['y_eval'].shape
    np.testing.assert_allclose(y.v, z['y_eval'], rtol=0, atol=2e-5)


@pytest.mark.parametrize('name', golden_names())
def test_train_forward_backward_matches_reference(name):
    cfg, z, state, grads, post = load_golden(name)
    m = _model(cfg)
    loss, y, g, buf = m.loss_and_grads(state, z['x'], z['y3d'], training=True)
    np.testing.assert_allclose(y, z['y_train'], rtol=0, atol=2e-5)
    assert abs(loss - float(z['loss'])) < 1e-5
    assert set(g) == set(grads)

    def worst(budget=None):
        w = ('', 0.0)
        for k in grads:
            scale = max(1e-3, float(np.abs(grads[k]).max()))
            e = np.abs(g[k] - grads[k])
            if budget is not None:
                e = np.maximum(e - 1.25 * budget[k], 0.0)
            if float(e.max()) / scale > w[1]:
                w = (k, float(e.max()) / scale)
        return w
    w = worst()
    if w[1] >= 2e-3:
        # a ReLU input within fp32 round-off of zero is undecidable for the fp32 reference: the oracle evaluates both decisions of
        # every |z| < 1e-6 and only the part of the difference they cannot explain counts (tests/parity_helpers.py::_tie_budget;
        # the 245-frame five-level fixture has 12 such inputs)
        from parity_helpers import _tie_budget
        n, budget = _tie_budget(lambda: m.loss_and_grads(state, z['x'], z['y3d'], training=True)[2], 1e-6)
        assert 0 < n <= 32, n
        w = worst(budget)
    assert w[1] < 2e-3, w
    for k in post:
        if k.endswith('num_batches_tracked'):
            assert int(buf[k]) == int(post[k])
        else:
            np.testing.assert_allclose(buf[k], post[k], rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize('name', golden_names())
def test_stock_torch_backend_matches_reference_and_numpy_oracle(name):
    """The same restatement on oracle.torch_ops (stock ATen operators, torch autograd; float64 on CPU here): pinned to the
    reference fixtures and to the numpy oracle, so it can serve as the GPU-side second reference / stock comparator."""
    import torch
    from oracle import torch_ops
    cfg, z, state, grads, post = load_golden(name)
    m = _model(cfg)
    loss_np, y_np, g_np, buf_np = m.loss_and_grads(state, z['x'], z['y3d'], training=True)
    tstate = {k: torch.from_numpy(np.array(v)) for k, v in state.items()}
    with go.use_backend(torch_ops):
        mt = _model(cfg, dtype=torch.float64)
        y_eval, _ = mt.forward(tstate, torch.from_numpy(z['x']), training=False)
        loss, y, g, buf = mt.loss_and_grads(tstate, torch.from_numpy(z['x']), torch.from_numpy(z['y3d']), training=True)
    np.testing.assert_allclose(y_eval.v.detach().numpy(), z['y_eval'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(y.detach().numpy(), y_np, rtol=0, atol=1e-9)
    assert abs(loss - loss_np) < 1e-10
    assert set(g) == set(g_np)
    for k in g_np:
        scale = max(1e-3, float(np.abs(g_np[k]).max()))
        assert float(np.abs(g[k].numpy() - g_np[k]).max()) / scale < 1e-8, k
    for k in post:
        np.testing.assert_allclose(np.asarray(buf[k]), buf_np[k], rtol=1e-10, atol=1e-12, err_msg=k)
    assert go.ag.__name__.endswith('np_autograd')      # the context manager restored the default backend
