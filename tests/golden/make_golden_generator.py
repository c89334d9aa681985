"""Generates tests/golden/generator_*.npz by running the REFERENCE ChunkedGenerator (reference common/generators.py) on small
synthetic sequences.  Run in the build container only (needs /root/reference): python tests/golden/make_golden_generator.py"""
import os, sys
import numpy as np
sys.path.insert(0, '/root/reference')
from common.generators import ChunkedGenerator, UnchunkedGenerator          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
KPS_LEFT, KPS_RIGHT = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]      # COCO (reference main.py keypoints_symmetry)
JOINTS_LEFT, JOINTS_RIGHT = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]            # Human3.6M skeleton (h36m_dataset.py)

CASES = {
    'train_like': dict(lengths=[50, 7, 33], batch_size=16, chunk_length=1, pad=13, causal_shift=0, shuffle=True, augment=True, cams=True),
    'causal_chunk3': dict(lengths=[20, 41], batch_size=5, chunk_length=3, pad=4, causal_shift=4, shuffle=True, augment=True, cams=False),
    'plain': dict(lengths=[9, 30], batch_size=8, chunk_length=1, pad=1, causal_shift=0, shuffle=False, augment=False, cams=False),
}

for name, c in CASES.items():
    rng = np.random.RandomState(7)
    # float32-representable values: the device path stores fp32, the comparison is then bit-exact
    p2 = [rng.randn(n, 17, 2).astype(np.float32).astype(np.float64) for n in c['lengths']]
    p3 = [rng.randn(n, 17, 3).astype(np.float32).astype(np.float64) for n in c['lengths']]
    cams = [rng.randn(9).astype(np.float32).astype(np.float64) for _ in c['lengths']] if c['cams'] else None
    gen = ChunkedGenerator(c['batch_size'], cams, p3, p2, c['chunk_length'], pad=c['pad'], causal_shift=c['causal_shift'],
                           shuffle=c['shuffle'], random_seed=1234, augment=c['augment'], kps_left=KPS_LEFT, kps_right=KPS_RIGHT,
                           joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT)
    out = {'lengths': np.array(c['lengths']), 'pairs_unshuffled': np.asarray(gen.pairs, dtype=np.int64)}
    for k, v in c.items():
        if k != 'lengths':
            out['cfg_' + k] = np.array(v)
    for i, s in enumerate(p2):
        out['p2_%d' % i] = s.astype(np.float32)
        out['p3_%d' % i] = p3[i].astype(np.float32)
        if cams is not None:
            out['cam_%d' % i] = cams[i]
    nb = 0
    for epoch in range(2):                       # two epochs: the shuffling stream continues across epochs
        for cam, b3, b2 in gen.next_epoch():
            if nb in (0, 1, gen.num_batches - 1):            # first two batches and the ragged last one
                out['e%d_b%d_2d' % (epoch, nb)] = b2.copy().astype(np.float32)
                out['e%d_b%d_3d' % (epoch, nb)] = b3.copy().astype(np.float32)
                if cam is not None:
                    out['e%d_b%d_cam' % (epoch, nb)] = cam.copy().astype(np.float32)
            nb += 1
        out['e%d_nbatches' % epoch] = np.array(nb)
        nb = 0
    np.savez_compressed(os.path.join(HERE, 'generator_%s.npz' % name), **out)
    print(name, 'pairs', len(gen.pairs), 'batches/epoch', gen.num_batches)

# ---- UnchunkedGenerator (evaluation side, reference common/generators.py:162-235): every sequence, plain + mirrored
rng = np.random.RandomState(11)
lengths = [5, 31]
p2 = [rng.randn(n, 17, 2).astype(np.float32).astype(np.float64) for n in lengths]
p3 = [rng.randn(n, 17, 3).astype(np.float32).astype(np.float64) for n in lengths]
cams = [rng.randn(9).astype(np.float32).astype(np.float64) for _ in lengths]
out = {'lengths': np.array(lengths)}
for i in range(len(lengths)):
    out['p2_%d' % i], out['p3_%d' % i], out['cam_%d' % i] = p2[i].astype(np.float32), p3[i].astype(np.float32), cams[i].astype(np.float32)
for tag, pad, cs, aug in (('sym_aug', 13, 0, True), ('causal_plain', 4, 4, False)):
    gen = UnchunkedGenerator(cams, p3, p2, pad=pad, causal_shift=cs, augment=aug, kps_left=KPS_LEFT, kps_right=KPS_RIGHT,
                             joints_left=JOINTS_LEFT, joints_right=JOINTS_RIGHT)
    for i, (cam, b3, b2) in enumerate(gen.next_epoch()):
        out['%s_%d_2d' % (tag, i)] = b2.astype(np.float32)
        out['%s_%d_3d' % (tag, i)] = b3.astype(np.float32)
        out['%s_%d_cam' % (tag, i)] = cam.astype(np.float32)
np.savez_compressed(os.path.join(HERE, 'unchunked_generator.npz'), **out)
print('unchunked ok')
