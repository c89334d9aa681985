"""Device-resident ChunkedGenerator against the reference class (golden fixtures) and the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

KPS_LEFT, KPS_RIGHT = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]
JOINTS_LEFT, JOINTS_RIGHT = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
CASES = sorted(os.path.basename(p)[len('generator_'):-4] for p in glob.glob(os.path.join(GOLDEN, 'generator_*.npz')))


def load(name):
    z = np.load(os.path.join(GOLDEN, 'generator_%s.npz' % name))
    n = len(z['lengths'])
    p2 = [z['p2_%d' % i].astype(np.float64) for i in range(n)]
    p3 = [z['p3_%d' % i].astype(np.float64) for i in range(n)]
    cams = [z['cam_%d' % i].astype(np.float64) for i in range(n)] if bool(z['cfg_cams']) else None
    cfg = dict(batch_size=int(z['cfg_batch_size']), chunk_length=int(z['cfg_chunk_length']), pad=int(z['cfg_pad']),
               causal_shift=int(z['cfg_causal_shift']), shuffle=bool(z['cfg_shuffle']), augment=bool(z['cfg_augment']))
    return z, p2, p3, cams, cfg


def make(cfg, p2, p3, cams, device):
    from gast_hip.generators import ChunkedGenerator
    return ChunkedGenerator(cfg['batch_size'], cams, p3, p2, cfg['chunk_length'], pad=cfg['pad'], causal_shift=cfg['causal_shift'],
                            shuffle=cfg['shuffle'], random_seed=1234, augment=cfg['augment'], kps_left=KPS_LEFT, kps_right=KPS_RIGHT,
                            joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT, device=device)


@pytest.mark.parametrize('name', CASES)
def test_pairs_table_shuffle_and_oracle_match_reference(name):
    """Host logic + oracle on CPU: the lineage table equals the reference's, the shuffled epoch order reproduces the reference's
    batches when fed to the numpy oracle (two epochs: the RandomState stream continues), the ragged last batch included."""
    from oracle.generators_oracle import build_batch
    z, p2, p3, cams, cfg = load(name)
    gen = make(cfg, p2, p3, cams, 'cpu')
    assert np.array_equal(np.asarray(gen.pairs, dtype=np.int64), z['pairs_unshuffled'])
    assert gen.num_batches == int(z['e0_nbatches']) and gen.num_frames() == gen.num_batches * cfg['batch_size']
    B = cfg['batch_size']
    for epoch in range(2):
        _, pairs = gen.next_pairs()
        table = gen.epoch_table(pairs)
        assert table.dtype == np.int32 and table.shape == (len(gen.pairs), 4)
        for b in (0, 1, gen.num_batches - 1):
            cam, b3, b2 = build_batch(table[b * B:(b + 1) * B], p2, p3, cams, cfg['chunk_length'], cfg['pad'], cfg['causal_shift'],
                                      KPS_LEFT, KPS_RIGHT, JOINTS_LEFT, JOINTS_RIGHT)
            assert np.array_equal(b2.astype(np.float32), z['e%d_b%d_2d' % (epoch, b)])
            assert np.array_equal(b3.astype(np.float32), z['e%d_b%d_3d' % (epoch, b)])
            if cams is not None:
                assert np.array_equal(cam.astype(np.float32), z['e%d_b%d_cam' % (epoch, b)])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        next(gen.next_epoch())


def test_flip_permutation():
    from gast_hip.generators import flip_permutation
    perm = flip_permutation(17, KPS_LEFT, KPS_RIGHT)
    x = np.arange(17)
    y = x.copy()
    y[KPS_LEFT + KPS_RIGHT] = x[KPS_RIGHT + KPS_LEFT]
    assert np.array_equal(x[perm], y)
    assert np.array_equal(flip_permutation(5, None, None), np.arange(5))


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_device_generator_matches_reference_bit_exact(name):
    """The HIP gather reproduces the reference generator's batches bit for bit (fp32), two epochs, ragged last batch."""
    z, p2, p3, cams, cfg = load(name)
    gen = make(cfg, p2, p3, cams, 'cuda')
    for epoch in range(2):
        nb = 0
        for cam, b3, b2 in gen.next_epoch():
            assert b2.is_cuda and b2.dtype == torch.float32
            if nb in (0, 1, gen.num_batches - 1):
                assert np.array_equal(b2.cpu().numpy(), z['e%d_b%d_2d' % (epoch, nb)]), (epoch, nb)
                assert np.array_equal(b3.cpu().numpy(), z['e%d_b%d_3d' % (epoch, nb)]), (epoch, nb)
                if cams is not None:
                    assert np.array_equal(cam.cpu().numpy(), z['e%d_b%d_cam' % (epoch, nb)]), (epoch, nb)
            nb += 1
        assert nb == gen.num_batches


@pytest.mark.gpu
def test_device_generator_feeds_the_model():
    """generator -> model -> mpjpe end to end on the device, shapes as in reference main.py:218-237 (pad = (RF-1)/2)."""
    from gast_hip.generators import ChunkedGenerator
    from gast_hip.loss import mpjpe
    from test_plan_cpu import build
    from tests_helpers import PARENTS
    rng = np.random.RandomState(0)
    p2 = [rng.randn(n, 17, 2) * 0.5 for n in (40, 25)]
    p3 = [rng.randn(n, 17, 3) * 0.3 for n in (40, 25)]
    m = build(dict(J=17, parents=PARENTS[17], arc=[3, 3], channels=16, causal=False, variant='strided')).cuda().train()
    pad = (m.receptive_field() - 1) // 2
    gen = ChunkedGenerator(32, None, p3, p2, 1, pad=pad, shuffle=True, augment=True, kps_left=KPS_LEFT, kps_right=KPS_RIGHT,
                           joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT)
    n = 0
    for _, b3, b2 in gen.next_epoch():
        pred = m(b2)
        assert pred.shape == b3.shape
        loss = mpjpe(pred, b3)
        loss.backward()
        assert torch.isfinite(loss)
        n += b2.shape[0]
    assert n == len(gen.pairs) == 2 * 65


def _load_unchunked():
    z = np.load(os.path.join(GOLDEN, 'unchunked_generator.npz'))
    n = len(z['lengths'])
    return (z, [z['p2_%d' % i].astype(np.float64) for i in range(n)], [z['p3_%d' % i].astype(np.float64) for i in range(n)],
            [z['cam_%d' % i].astype(np.float64) for i in range(n)])


def test_unchunked_oracle_matches_reference():
    """the same gather restated in numpy (oracle) reproduces the reference UnchunkedGenerator: a sequence is one chunk of its own
    length, the mirrored copy is the `flip` row"""
    from oracle.generators_oracle import build_batch
    z, p2, p3, cams = _load_unchunked()
    for tag, pad, cs, aug in (('sym_aug', 13, 0, True), ('causal_plain', 4, 4, False)):
        for i, n in enumerate(z['lengths']):
            rows = [(i, 0, n, 0)] + ([(i, 0, n, 1)] if aug else [])
            cam, b3, b2 = build_batch(rows, p2, p3, cams, int(n), pad, cs, KPS_LEFT, KPS_RIGHT, JOINTS_LEFT, JOINTS_RIGHT)
            assert np.array_equal(b2.astype(np.float32), z['%s_%d_2d' % (tag, i)])
            assert np.array_equal(b3.astype(np.float32), z['%s_%d_3d' % (tag, i)])
            assert np.array_equal(cam.astype(np.float32), z['%s_%d_cam' % (tag, i)])


@pytest.mark.gpu
def test_device_unchunked_generator_bit_exact():
    from gast_hip.generators import UnchunkedGenerator
    z, p2, p3, cams = _load_unchunked()
    for tag, pad, cs, aug in (('sym_aug', 13, 0, True), ('causal_plain', 4, 4, False)):
        gen = UnchunkedGenerator(cams, p3, p2, pad=pad, causal_shift=cs, augment=aug, kps_left=KPS_LEFT, kps_right=KPS_RIGHT,
                                 joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT, device='cuda')
        assert gen.num_frames() == int(z['lengths'].sum()) and gen.augment_enabled() == aug
        for i, (cam, b3, b2) in enumerate(gen.next_epoch()):
            assert np.array_equal(b2.cpu().numpy(), z['%s_%d_2d' % (tag, i)]), (tag, i)
            assert np.array_equal(b3.cpu().numpy(), z['%s_%d_3d' % (tag, i)]), (tag, i)
            assert np.array_equal(cam.cpu().numpy(), z['%s_%d_cam' % (tag, i)]), (tag, i)
