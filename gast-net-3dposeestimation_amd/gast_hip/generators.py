"""Device-resident drop-in for the reference's `common.generators.ChunkedGenerator` (reference common/generators.py:5-159;
SURVEY.md section 8 row f2).

Import it as `from gast_hip.generators import ChunkedGenerator` in place of `from common.generators import ChunkedGenerator`
(it deliberately does NOT live in a package called `common`: that would shadow the reference's own `common/` on sys.path).
Same constructor arguments, same `pairs` lineage table, same shuffling stream (`np.random.RandomState(random_seed).permutation`),
same batches in the same order -- but the sequences are uploaded ONCE (concatenated, fp32), every epoch's shuffled pair table is
uploaded once, and each batch is built by one HIP gather launch (`gast_chunk_gather`, csrc/data_ops.hip).  `next_epoch()` yields
`(batch_cam, batch_3d, batch_2d)` as float32 **device tensors** (the reference yields float64 numpy buffers that main.py:218-226
converts with `torch.from_numpy(x.astype('float32')).cuda()`; that line becomes a no-op).  The yielded tensors are reused buffers,
exactly like the reference's `self.batch_2d`: consume a batch before asking for the next one.

There is no CPU implementation of the gather: `next_epoch()` raises unless the generator lives on a GPU.
"""
import numpy as np
import torch


def build_pairs(lengths, chunk_length, augment):
    """The lineage table of generators.py:31-42: (seq_idx, start_frame, end_frame, flip) rows, un-augmented rows of a sequence
    first, then (when `augment`) the same rows flagged for flipping."""
    pairs = []
    for i, n in enumerate(lengths):
        n_chunks = (n + chunk_length - 1) // chunk_length
        offset = (n_chunks * chunk_length - n) // 2
        bounds = np.arange(n_chunks + 1) * chunk_length - offset
        idx = np.repeat(i, len(bounds) - 1)
        flags = np.full(len(bounds) - 1, False, dtype=bool)
        pairs += zip(idx, bounds[:-1], bounds[1:], flags)
        if augment:
            pairs += zip(idx, bounds[:-1], bounds[1:], ~flags)
    return pairs


def flip_permutation(n_joints, left, right):
    """perm[j] = source joint of destination joint j under `x[:, left + right] = x[:, right + left]` (generators.py:114-115)."""
    perm = np.arange(n_joints, dtype=np.int32)
    left, right = list(left or []), list(right or [])
    for dst, src in zip(left + right, right + left):
        perm[dst] = src
    return perm


class ChunkedGenerator:
    """
        Batched data generator, used for training (device-resident twin of the reference class; same arguments).
        `device`: where the sequences live (default: the current CUDA device).
    """
    def __init__(self, batch_size, cameras, poses_3d, poses_2d,
                 chunk_length, pad=0, causal_shift=0,
                 shuffle=True, random_seed=1234,
                 augment=False, kps_left=None, kps_right=None, joints_left=None, joints_right=None,
                 endless=False, device=None):
        assert poses_3d is None or len(poses_3d) == len(poses_2d), (len(poses_3d), len(poses_2d))
        assert cameras is None or len(cameras) == len(poses_2d)
        for i in range(len(poses_2d)):
            assert poses_3d is None or poses_3d[i].shape[0] == poses_2d[i].shape[0]
        lengths = [p.shape[0] for p in poses_2d]
        self.pairs = build_pairs(lengths, chunk_length, augment)
        self.num_batches = (len(self.pairs) + batch_size - 1) // batch_size
        self.batch_size = batch_size
        self.random = np.random.RandomState(random_seed)
        self.shuffle = shuffle
        self.pad = pad
        self.causal_shift = causal_shift
        self.endless = endless
        self.state = None
        self.chunk_length = chunk_length

        self.augment = augment
        self.kps_left = kps_left
        self.kps_right = kps_right
        self.joints_left = joints_left
        self.joints_right = joints_right

        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        self.device = torch.device(device)
        f32 = lambda seqs: torch.from_numpy(np.ascontiguousarray(np.concatenate(seqs, axis=0), dtype=np.float32)).to(self.device)  # noqa: E731
        self.poses_2d = f32(poses_2d)                                  # [sum_len][J2][F2]
        self.poses_3d = f32(poses_3d) if poses_3d is not None else None
        self.cameras = (torch.from_numpy(np.ascontiguousarray(np.stack(cameras), dtype=np.float32)).to(self.device)
                        if cameras is not None else None)
        self.seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)).to(self.device)
        J2 = self.poses_2d.shape[1]
        self.perm_2d = torch.from_numpy(flip_permutation(J2, kps_left, kps_right)).to(self.device)
        self.perm_3d = (torch.from_numpy(flip_permutation(self.poses_3d.shape[1], joints_left, joints_right)).to(self.device)
                        if self.poses_3d is not None else None)
        B = batch_size
        self.batch_2d = torch.empty(B, chunk_length + 2 * pad, J2, self.poses_2d.shape[2], dtype=torch.float32, device=self.device)
        self.batch_3d = (torch.empty(B, chunk_length, self.poses_3d.shape[1], self.poses_3d.shape[2], dtype=torch.float32,
                                     device=self.device) if self.poses_3d is not None else None)
        self.batch_cam = (torch.empty(B, self.cameras.shape[-1], dtype=torch.float32, device=self.device)
                          if self.cameras is not None else None)
        self._ops = None

    def num_frames(self):
        return self.num_batches * self.batch_size

    def random_state(self):
        return self.random

    def set_random_state(self, random):
        self.random = random

    def augment_enabled(self):
        return self.augment

    def next_pairs(self):
        if self.state is None:
            if self.shuffle:
                pairs = self.random.permutation(self.pairs)
            else:
                pairs = self.pairs
            return 0, pairs
        else:
            return self.state

    def epoch_table(self, pairs):
        """(N, 4) int32 host table of an epoch's pair order (what gets uploaded)."""
        return np.ascontiguousarray(np.asarray(pairs, dtype=np.int64).astype(np.int32).reshape(-1, 4))

    def next_epoch(self):
        if self.device.type != 'cuda':
            raise RuntimeError('ChunkedGenerator: sequences are on %s; the gather is a HIP kernel (no CPU fallback)' % self.device)
        if self._ops is None:
            from gast_hip.binding import HipOps
            self._ops = HipOps()
        enabled = True
        while enabled:
            start_idx, pairs = self.next_pairs()
            table = torch.from_numpy(self.epoch_table(pairs)).to(self.device)       # one upload per epoch
            n = table.shape[0]
            for b_i in range(start_idx, self.num_batches):
                first = b_i * self.batch_size
                nb = min(self.batch_size, n - first)
                self._ops.chunk_gather(self.poses_2d, self.poses_3d, self.cameras, self.seq_off, table, first, nb, self.chunk_length,
                                       self.pad, self.causal_shift, self.perm_2d, self.perm_3d, self.batch_2d, self.batch_3d,
                                       self.batch_cam)
                if self.endless:
                    self.state = (b_i + 1, pairs)
                cam = self.batch_cam[:nb] if self.batch_cam is not None else None
                b3 = self.batch_3d[:nb] if self.batch_3d is not None else None
                yield cam, b3, self.batch_2d[:nb]

            if self.endless:
                self.state = None
            else:
                enabled = False


class UnchunkedGenerator:
    """
    Non-batched data generator, used for testing (device-resident twin of reference common/generators.py:162-235; same arguments
    + `device`).  Sequences are returned one at a time, un-chunked, edge-padded by `pad` (shifted by `causal_shift`); with
    `augment` the batch holds the sequence and its mirrored copy (test-time augmentation, reference main.py:313-318).  Yields
    float32 device tensors built by one `gast_chunk_gather` launch per sequence.
    """

    def __init__(self, cameras, poses_3d, poses_2d, pad=0, causal_shift=0,
                 augment=False, kps_left=None, kps_right=None, joints_left=None, joints_right=None, device=None):
        assert poses_3d is None or len(poses_3d) == len(poses_2d)
        assert cameras is None or len(cameras) == len(poses_2d)
        self.augment = augment
        self.kps_left = kps_left
        self.kps_right = kps_right
        self.joints_left = joints_left
        self.joints_right = joints_right
        self.pad = pad
        self.causal_shift = causal_shift
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        self.device = torch.device(device)
        self.lengths = [p.shape[0] for p in poses_2d]
        f32 = lambda seqs: torch.from_numpy(np.ascontiguousarray(np.concatenate(seqs, axis=0), dtype=np.float32)).to(self.device)  # noqa: E731
        self.poses_2d = f32(poses_2d)
        self.poses_3d = f32(poses_3d) if poses_3d else None
        self.cameras = (torch.from_numpy(np.ascontiguousarray(np.stack(cameras), dtype=np.float32)).to(self.device) if cameras else None)
        self.seq_off = torch.from_numpy(np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)).to(self.device)
        self.perm_2d = torch.from_numpy(flip_permutation(self.poses_2d.shape[1], kps_left, kps_right)).to(self.device)
        self.perm_3d = (torch.from_numpy(flip_permutation(self.poses_3d.shape[1], joints_left, joints_right)).to(self.device)
                        if self.poses_3d is not None else None)
        # (seq, start, end, flip) rows: the plain and the mirrored copy of every sequence
        rows = []
        for i, n in enumerate(self.lengths):
            rows += [(i, 0, n, 0), (i, 0, n, 1)]
        self.table = torch.from_numpy(np.asarray(rows, dtype=np.int32)).to(self.device)
        self._ops = None

    def num_frames(self):
        return sum(self.lengths)

    def augment_enabled(self):
        return self.augment

    def set_augment(self, augment):
        self.augment = augment

    def next_epoch(self):
        if self.device.type != 'cuda':
            raise RuntimeError('UnchunkedGenerator: sequences are on %s; the gather is a HIP kernel (no CPU fallback)' % self.device)
        if self._ops is None:
            from gast_hip.binding import HipOps
            self._ops = HipOps()
        nb = 2 if self.augment else 1
        for i, n in enumerate(self.lengths):
            b2 = torch.empty(nb, n + 2 * self.pad, self.poses_2d.shape[1], self.poses_2d.shape[2], dtype=torch.float32, device=self.device)
            b3 = (torch.empty(nb, n, self.poses_3d.shape[1], self.poses_3d.shape[2], dtype=torch.float32, device=self.device)
                  if self.poses_3d is not None else None)
            cam = (torch.empty(nb, self.cameras.shape[-1], dtype=torch.float32, device=self.device) if self.cameras is not None else None)
            self._ops.chunk_gather(self.poses_2d, self.poses_3d, self.cameras, self.seq_off, self.table, 2 * i, nb, n, self.pad,
                                   self.causal_shift, self.perm_2d, self.perm_3d, b2, b3, cam)
            yield cam, b3, b2
