"""The reference-caller scenario shared by tests/caller/run_reference_caller.py (which drives the reference's own main.py) and
tests/test_reference_caller.py (GPU: the same steps restated, because /root/reference does not exist on the GPU box).

Scenario = what trainval.py does, shrunk: create_model (main.py:150-187) -> one epoch of train() (main.py:213-243) with
Adam(amsgrad) as trainval.py:78 -> model_pos.load_state_dict(model_pos_train.state_dict()) (trainval.py:124) -> evaluate() with
flip test-time augmentation (main.py:299-353) -> evaluate(return_predictions=True) on the first sequence."""
import numpy as np

PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15]          # reference reconstruction.py:95 / h36m after joint removal
JOINTS_LEFT, JOINTS_RIGHT = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]      # reference gen_skes.py:36
KPS_LEFT, KPS_RIGHT = JOINTS_LEFT, JOINTS_RIGHT
ARCH, CHANNELS, BATCH, LR = '3,3', 16, 16, 1e-3
# two sizes: 'short' = 2 optimizer steps (tight comparison), 'epoch' = 7 steps.  Parameters with an analytically ZERO gradient (a bias
# in front of a BatchNorm: init_bn.bias, the attention's g / theta / phi biases) receive round-off noise as gradient, which Adam
# normalises into steps of the size of the learning rate in an implementation-dependent direction; the network's outputs are
# invariant to them, so the comparisons are made on losses, errors and predictions, and on the other parameters.
LENGTHS = {'short': [12], 'epoch': [37, 41, 29]}
ZERO_GRAD_PARAMS = ('init_bn.bias', '.g.bias', '.theta.bias', '.phi.bias')


def make_data(size):
    rng = np.random.RandomState(20260926)
    lengths = LENGTHS[size]
    poses_2d = [rng.uniform(-1, 1, size=(n, 17, 2)).astype(np.float32) for n in lengths]
    poses_3d = []
    for n in lengths:
        p = (rng.randn(n, 17, 3) * 0.3).astype(np.float32)
        p[:, 0] = 0
        poses_3d.append(p)
    cameras = [rng.uniform(-1, 1, size=9).astype(np.float32) for _ in lengths]
    return cameras, poses_3d, poses_2d
