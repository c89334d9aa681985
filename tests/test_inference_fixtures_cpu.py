"""The reference-generated inference fixtures (tests/golden/make_golden_inference.py), CPU side:
(a) the drop-in's seed-constructed models carry exactly the weights the reference had when the fixtures were made (digest) -- the
premise of tests/test_inference_gpu.py; (b) the numpy oracle is pinned to the clip / 243-frame fixtures as well."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from tests_helpers import PARENTS, INFERENCE_CASES, SHAPE243, perturb_like_golden, state_digest

KPS_LEFT, KPS_RIGHT = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]


def build_case(case, dropout=None):
    from model.gast_net import SpatioTemporalModel, SpatioTemporalModelOptimized1f
    from oracle.gast_oracle import adj_from_parents
    adj = torch.from_numpy(adj_from_parents(PARENTS[17]))
    torch.manual_seed(case['seed'])
    if case.get('cls') == 'strided':
        m = SpatioTemporalModelOptimized1f(adj, 17, 2, 17, filter_widths=case['arc'], causal=case['causal'], channels=case['channels'],
                                           dropout=0.25 if dropout is None else dropout)
    else:
        m = SpatioTemporalModel(adj, 17, 2, 17, filter_widths=case['arc'], causal=case.get('causal', False), channels=case['channels'],
                                dropout=0.05 if dropout is None else dropout)
    return m


def padded_clip(kpts, pad, shift, flip):
    """reference common/generators.py:214-235 (UnchunkedGenerator.next_epoch): edge padding by (pad + shift, pad - shift), the
    mirrored copy as a second batch entry"""
    x = np.pad(kpts, ((pad + shift, pad - shift), (0, 0), (0, 0)), 'edge')[None]
    if flip:
        x = np.concatenate([x, x], axis=0)
        x[1, :, :, 0] *= -1
        x[1, :, KPS_LEFT + KPS_RIGHT] = x[1, :, KPS_RIGHT + KPS_LEFT]
    return x.astype(np.float32)


def unflip_mean(pred):
    """reference main.py:313-318 / reconstruction.py:163-167 on an ndarray (2, T, J, 3)"""
    pred = pred.copy()
    pred[1, :, :, 0] *= -1
    pred[1, :, KPS_LEFT + KPS_RIGHT] = pred[1, :, KPS_RIGHT + KPS_LEFT]
    return pred.mean(axis=0)


@pytest.mark.parametrize('name', list(INFERENCE_CASES))
def test_seed_constructed_weights_are_the_fixture_weights(name):
    case = INFERENCE_CASES[name]
    z = np.load(os.path.join(GOLDEN, 'inf_%s.npz' % name))
    m = build_case(case)
    perturb_like_golden(m, torch.Generator().manual_seed(case['seed'] + 1))
    assert state_digest(m.state_dict()) == str(z['digest'])
    assert z['kpts'].shape == (277, 17, 2) and z['pred'].shape == (277, 17, 3)


def test_shape243_seed_constructed_weights():
    z = np.load(os.path.join(GOLDEN, 'shape243_j17_c32.npz'))
    m = build_case(dict(SHAPE243, causal=False), dropout=0.0)
    perturb_like_golden(m, torch.Generator().manual_seed(SHAPE243['seed'] + 1))
    assert state_digest(m.state_dict()) == str(z['digest'])
    assert sum(p.numel() for p in m.parameters()) == 7091736            # arc 3^5, C0 = 32
    assert m.receptive_field() == 243


@pytest.mark.parametrize('name,frames', [('baseball_sym243', None), ('baseball_causal27_dil', 48)])
def test_oracle_reproduces_the_clip_predictions(name, frames):
    """The oracle restatement (float64, stock ATen backend for speed: oracle/torch_ops.py is pinned to the numpy oracle at 1e-9 by
    tests/test_oracle_golden.py) over the edge-padded clip against the reference's output.  The causal model's frame t depends on
    frames <= t only, so its first 48 frames + mirrored copy suffice (flip TTA checked there); the symmetric 243-frame model needs
    the whole clip (un-mirrored copy only, to keep the CPU suite short)."""
    from oracle import gast_oracle as go
    from oracle import torch_ops
    case = INFERENCE_CASES[name]
    z = np.load(os.path.join(GOLDEN, 'inf_%s.npz' % name))
    m = build_case(case)
    perturb_like_golden(m, torch.Generator().manual_seed(case['seed'] + 1))
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    pad = int(z['pad'])
    with go.use_backend(torch_ops):
        om = go.OracleModel(go.adj_from_parents(PARENTS[17]), case['arc'], case['channels'], causal=case['causal'], variant='dilated',
                            dtype=torch.float64)
        if frames is None:
            x = padded_clip(z['kpts'], pad, 0, False)
            with torch.no_grad():
                y, _ = om.forward(state, torch.from_numpy(x), training=False)
            np.testing.assert_allclose(y.v.numpy()[0], z['pred_noflip'], rtol=0, atol=2e-5)
        else:
            x = padded_clip(z['kpts'], pad, pad, True)[:, :frames + 2 * pad]
            with torch.no_grad():
                y, _ = om.forward(state, torch.from_numpy(x), training=False)
            y = y.v.numpy()
            np.testing.assert_allclose(unflip_mean(y), z['pred'][:frames], rtol=0, atol=2e-5)
            np.testing.assert_allclose(y[0], z['pred_noflip'][:frames], rtol=0, atol=2e-5)
