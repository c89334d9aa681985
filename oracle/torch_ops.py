"""Stock-PyTorch backend for the oracle -- TEST / BASELINE INFRASTRUCTURE ONLY.

Same primitive names as `np_autograd`, implemented with the stock ATen operators the reference itself issues
(`F.conv2d`, `F.conv1d`, `F.batch_norm`, `torch.matmul`, `softmax`, `cat`, boolean-mask assignment) and
torch's own autograd, on whatever device the state tensors live on.  `gast_oracle.use_backend(torch_ops)` runs the very same
restatement of the model through it, which gives

  * the "stock PyTorch-ROCm" comparator of SURVEY.md section 8(d): what a user of the reference gets on an MI355X by calling
    `model.cuda()` (MIOpen / rocBLAS / hipBLASLt kernels, one launch per ATen op), timed by `bench.py --stock-baseline`;
  * a second, GPU-side parity reference at BASELINE.json's full size (tests/test_model_gpu.py), pinned to the numpy oracle --
    and through it to the reference's golden fixtures -- by tests/test_oracle_golden.py on CPU.

Like the rest of `oracle/` it is imported by tests/, `__graft_entry__.smoke()` and bench.py's baseline legs only; the product
path (gast-net-3dposeestimation_amd/) never imports it.
"""
import numpy as np
import torch
import torch.nn.functional as F


class Var:
    """A tensor with the two attributes the restatement reads: `.v` (value) and, after backward(), `.g` (gradient)."""
    __slots__ = ('v',)

    def __init__(self, v):
        self.v = v

    @property
    def shape(self):
        return tuple(self.v.shape)

    @property
    def g(self):
        return self.v.grad


_DT = {np.float32: torch.float32, np.float64: torch.float64, torch.float32: torch.float32, torch.float64: torch.float64}


def asarray(v, dtype, like=None):
    """Value -> tensor of `dtype` on the device of `like` (or of `v` when it already is a tensor)."""
    dt = _DT.get(dtype, dtype)
    if torch.is_tensor(v):
        return v.detach().to(dt).clone()
    t = torch.as_tensor(np.asarray(v))
    if t.is_floating_point():
        t = t.to(dt)
    return t.to(like.device) if like is not None else t


def leaf(v, needs=True):
    t = v.detach().clone()
    if needs and t.is_floating_point():
        t.requires_grad_(True)
    return Var(t)


def const(v):
    return Var(v.detach())


def eye(J, like):
    return torch.eye(J, dtype=like.v.dtype, device=like.v.device)[None]


def backward(root, seed=None):
    if seed is None:
        root.v.backward()
    else:
        root.v.backward(torch.as_tensor(seed, dtype=root.v.dtype, device=root.v.device))


def add(a, b):
    return Var(a.v + b.v)


def mul_const(a, c):
    c = c if torch.is_tensor(c) else torch.as_tensor(np.asarray(c), dtype=a.v.dtype, device=a.v.device)
    return Var(a.v * c)


def _forced(v):
    """The forced-decision hook of oracle/np_autograd.py (TIES['forced']) for this backend: None, or the boolean tensor of this call."""
    from oracle.np_autograd import TIES
    forced = TIES.get('forced')
    if forced is None:
        return None
    m = forced[TIES['pos']]
    TIES['pos'] += 1
    m = torch.as_tensor(m, device=v.device).bool()
    assert m.shape == v.shape, ('forced decision %d has shape %s, the call sees %s' % (TIES['pos'] - 1, tuple(m.shape), tuple(v.shape)))
    dis = m != (v.detach() > 0)
    n = int(dis.sum())
    if n:
        TIES['flips'] += n
        TIES['flip_max'] = max(TIES['flip_max'], float(v.detach()[dis].abs().max()))
    return m


def relu(a):
    m = _forced(a.v)
    if m is not None:
        return Var(torch.where(m, a.v, torch.zeros((), dtype=a.v.dtype, device=a.v.device)))
    return Var(torch.relu(a.v))


def leaky_relu(a, slope):
    m = _forced(a.v)
    if m is not None:
        return Var(torch.where(m, a.v, a.v * slope))
    return Var(F.leaky_relu(a.v, slope))


def permute(a, axes):
    return Var(a.v.permute(*axes))


def reshape(a, shape):
    return Var(a.v.reshape(shape))


def getitem(a, idx):
    return Var(a.v[idx])


def cat(vs, axis):
    return Var(torch.cat([v.v for v in vs], dim=axis))


def matmul(a, b):
    return Var(torch.matmul(a.v, b.v))


def softmax(a, axis=-1):
    return Var(torch.softmax(a.v, dim=axis))


def masked_fill_from(e, mask, fill):
    m = torch.as_tensor(np.ascontiguousarray(mask), device=e.v.device)
    v = torch.full(tuple(mask.shape), fill, dtype=e.v.dtype, device=e.v.device)
    v[m] = e.v.reshape(-1)
    return Var(v)


def conv2d_k1(x, w, dilation=1, stride=1):
    return Var(F.conv2d(x.v, w.v, stride=(stride, 1), dilation=(dilation, 1)))


def conv1d_1x1(x, w, b):
    return Var(F.conv1d(x.v, w.v, b.v))


def batch_norm2d(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    return Var(F.batch_norm(x.v, running_mean, running_var, gamma.v, beta.v, training, momentum, eps))


def dropout(a, p):
    return Var(F.dropout(a.v, p, True))


def mpjpe(pred, target):
    t = target if torch.is_tensor(target) else torch.as_tensor(np.asarray(target), dtype=pred.v.dtype, device=pred.v.device)
    return Var(torch.mean(torch.norm(pred.v - t, dim=-1)))
