"""TEST INFRASTRUCTURE (imported only by tests/): numpy restatement of the reference's ChunkedGenerator batch construction
(reference common/generators.py:93-159), one batch at a time from an explicit pair table.  Pinned against the reference class
itself by tests/golden/make_golden_generator.py -> tests/golden/generator_*.npz (tests/test_generators.py)."""
import numpy as np


def build_batch(pairs_rows, poses_2d, poses_3d, cameras, chunk_length, pad, causal_shift, kps_left, kps_right, joints_left, joints_right):
    """pairs_rows: iterable of (seq_i, start_3d, end_3d, flip).  Returns (batch_cam | None, batch_3d | None, batch_2d) float64."""
    rows = list(pairs_rows)
    B = len(rows)
    b2 = np.empty((B, chunk_length + 2 * pad) + poses_2d[0].shape[1:])
    b3 = np.empty((B, chunk_length) + poses_3d[0].shape[1:]) if poses_3d is not None else None
    bc = np.empty((B, cameras[0].shape[-1])) if cameras is not None else None
    for i, (seq_i, start_3d, end_3d, flip) in enumerate(rows):
        seq_i, start_3d, end_3d = int(seq_i), int(start_3d), int(end_3d)
        start_2d = start_3d - pad - causal_shift                      # generators.py:97-98
        end_2d = end_3d + pad - causal_shift
        seq_2d = poses_2d[seq_i]
        low, high = max(start_2d, 0), min(end_2d, seq_2d.shape[0])    # generators.py:102-110
        pl, pr = low - start_2d, end_2d - high
        b2[i] = np.pad(seq_2d[low:high], ((pl, pr), (0, 0), (0, 0)), 'edge') if (pl or pr) else seq_2d[low:high]
        if flip:                                                      # generators.py:112-115
            b2[i, :, :, 0] *= -1
            b2[i, :, kps_left + kps_right] = b2[i, :, kps_right + kps_left]
        if poses_3d is not None:                                      # generators.py:118-133
            seq_3d = poses_3d[seq_i]
            low, high = max(start_3d, 0), min(end_3d, seq_3d.shape[0])
            pl, pr = low - start_3d, end_3d - high
            b3[i] = np.pad(seq_3d[low:high], ((pl, pr), (0, 0), (0, 0)), 'edge') if (pl or pr) else seq_3d[low:high]
            if flip:
                b3[i, :, :, 0] *= -1
                b3[i, :, joints_left + joints_right] = b3[i, :, joints_right + joints_left]
        if cameras is not None:                                       # generators.py:136-141
            bc[i] = cameras[seq_i]
            if flip:
                bc[i, 2] *= -1
                bc[i, 7] *= -1
    return bc, b3, b2
