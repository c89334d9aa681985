"""Causal streaming inference (gast_hip/streaming.py): one frame per call through per-level frame
buffers must reproduce the window forward of the causal model on the edge-padded clip (reference gen_skes.py:43-69,
tools/inference.py:73-91), for both interchangeable variants, with and without flip test-time augmentation."""
import numpy as np
import pytest
import torch

from tests_helpers import PARENTS

pytestmark = pytest.mark.gpu


def _adj(J):
    a = np.zeros((J, J))
    for i, p in enumerate(PARENTS[J]):
        if p >= 0:
            a[i, p] = a[p, i] = 1.0
    a += np.eye(J)
    return torch.from_numpy((a / a.sum(1, keepdims=True)).astype(np.float32))


def _model(cls, arc, ch, J=17, seed=3):
    torch.manual_seed(seed)
    m = cls(_adj(J), J, 2, J, filter_widths=list(arc), causal=True, dropout=0.25, channels=ch)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, b in m.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif n.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        for n, p in m.named_parameters():
            if n.endswith('C_k'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    return m.cuda().eval()


def _window_forward(model, clip):
    """the reference's evaluation of a causal model: left edge padding by receptive_field - 1, one window forward"""
    rf = model.receptive_field()
    padded = torch.cat([clip[:, :1].expand(-1, rf - 1, -1, -1), clip], dim=1)
    with torch.no_grad():
        return model(padded.contiguous())


@pytest.mark.parametrize('arc,ch,T', [((3, 3, 3), 32, 300), ((3, 3, 3, 3), 16, 100), ((5, 3), 16, 40)])
@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'hipgraph'])
def test_stream_equals_window_forward(arc, ch, T, graph):
    from model.gast_net import SpatioTemporalModel
    from gast_hip.streaming import CausalStream
    m = _model(SpatioTemporalModel, arc, ch)
    g = torch.Generator().manual_seed(11)
    clip = (torch.rand(2, T, 17, 2, generator=g) * 2 - 1).cuda()        # (300-frame clip: the shape of data/keypoints/baseball.json x 2)
    ref = _window_forward(m, clip)
    out = CausalStream(m, batch=2, graph=graph).run(clip)
    assert out.shape == ref.shape == (2, T, 17, 3)
    err = float((out - ref).abs().max())
    # (same kernels; the one-frame GEMMs take the split-K path, i.e. another summation order: not bit-equal, fp32 round-off)
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), 'stream vs window forward: max err %.3e' % err


def test_stream_accepts_the_strided_twin_and_flip_tta():
    from model.gast_net import SpatioTemporalModel, SpatioTemporalModelOptimized1f
    from gast_hip.streaming import CausalStream
    md = _model(SpatioTemporalModel, (3, 3, 3), 32)
    ms = _model(SpatioTemporalModelOptimized1f, (3, 3, 3), 32)
    ms.load_state_dict(md.state_dict())                                    # interchangeable weights (reference gast_net.py:180-251)
    g = torch.Generator().manual_seed(12)
    clip = (torch.rand(1, 50, 17, 2, generator=g) * 2 - 1).cuda()
    kl, kr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
    # reference main.py:313-318 / tools/inference.py:80-84: mirrored copy, un-mirror, average
    flipped = clip.clone()
    flipped[..., 0] *= -1
    flipped[:, :, kl + kr] = flipped[:, :, kr + kl]
    ref = _window_forward(md, torch.cat([clip, flipped], dim=0))
    ref[1, :, :, 0] *= -1
    ref[1, :, kl + kr] = ref[1, :, kr + kl]
    ref = ref.mean(dim=0, keepdim=True)
    out = CausalStream(ms, batch=1, flip=(kl, kr, kl, kr)).run(clip)
    err = float((out - ref).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), 'flip-TTA stream vs window forward: max err %.3e' % err


def test_stream_rejects_symmetric_models():
    from model.gast_net import SpatioTemporalModel
    from gast_hip.streaming import CausalStream
    torch.manual_seed(0)
    m = SpatioTemporalModel(_adj(17), 17, 2, 17, filter_widths=[3, 3], causal=False, channels=16).cuda().eval()
    with pytest.raises(ValueError, match='causal'):
        CausalStream(m)
