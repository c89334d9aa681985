"""Tiny reverse-mode autodiff over numpy arrays -- TEST INFRASTRUCTURE ONLY.

This file is part of the parity oracle (see oracle/README.md).  It is imported by tests/, by
`__graft_entry__.smoke()` and by `bench.py`'s `cpu_baseline` leg, and by nothing else: the product path
(gast-net-3dposeestimation_amd/) never imports it and has no CPU fallback.

It provides just the primitives the reference's hot path issues through ATen:
matmul, conv2d with a (k,1) kernel / dilation / stride, batch_norm, relu, leaky_relu, softmax, cat, slicing,
permute/reshape, the boolean-mask scatter `adj[m] = e` and the mean-L2 loss.  Each primitive carries its own
hand-written vector-Jacobian product; tests/test_oracle_autograd.py checks them by finite differences.
"""
import numpy as np


class Var:
    """A node of the tape: value, gradient slot, parents and a closure that pushes the gradient to the parents."""
    __slots__ = ('v', 'g', 'parents', 'bw', 'needs')

    def __init__(self, v, parents=(), bw=None, needs=None):
        self.v = v
        self.g = None
        self.parents = parents
        self.bw = bw
        self.needs = any(p.needs for p in parents) if needs is None else needs

    @property
    def shape(self):
        return self.v.shape

    def acc(self, g):
        if not self.needs:
            return
        if self.g is None:
            self.g = np.array(g, dtype=self.v.dtype, copy=True)
        else:
            self.g += g


def leaf(v, needs=True):
    return Var(np.asarray(v), (), None, needs)


def const(v):
    return Var(np.asarray(v), (), None, False)


def asarray(v, dtype, like=None):
    """A private copy of `v` as an ndarray of `dtype` (None: keep).  (Backend hook: oracle/torch_ops.py has the tensor twin.)"""
    return np.array(v) if dtype is None else np.array(v, dtype=dtype)


def eye(J, like):
    return np.eye(J, dtype=like.v.dtype)[None]


def backward(root, seed=None):
    order, seen = [], set()

    def visit(n):
        stack = [(n, False)]
        while stack:
            node, done = stack.pop()
            if done:
                order.append(node)
                continue
            if id(node) in seen:
                continue
            seen.add(id(node))
            stack.append((node, True))
            for p in node.parents:
                if id(p) not in seen and p.needs:
                    stack.append((p, False))
    visit(root)
    root.g = np.ones_like(root.v) if seed is None else np.asarray(seed, dtype=root.v.dtype)
    for node in reversed(order):
        if node.bw is not None and node.g is not None:
            node.bw(node.g)


def _unbroadcast(g, shape):
    while g.ndim > len(shape):
        g = g.sum(axis=0)
    for ax, (gs, s) in enumerate(zip(g.shape, shape)):
        if s == 1 and gs != 1:
            g = g.sum(axis=ax, keepdims=True)
    return g


# ---------------------------------------------------------------- elementwise / shape ops
def add(a, b):
    out = Var(a.v + b.v, (a, b))

    def bw(g):
        a.acc(_unbroadcast(g, a.v.shape))
        b.acc(_unbroadcast(g, b.v.shape))
    out.bw = bw
    return out


def mul_const(a, c):
    c = np.asarray(c)
    out = Var(a.v * c, (a,))
    out.bw = lambda g: a.acc(_unbroadcast(g * c, a.v.shape))
    return out


# Tie handling for parity tests.  A ReLU / LeakyReLU input closer to zero than the round-off of the path under test is
# undecidable for it: whichever side it lands on, the gradient changes by that element's whole contribution.  Two tools:
#  * TIES['eps'] > 0: inputs with |v| < eps are decided by TIES['side'] ('on': treated as positive, 'off': as negative) and
#    counted, so a test can evaluate both decisions and budget the difference (tests/parity_helpers.py::_tie_budget);
#  * TIES['forced'] = [mask, ...]: the decisions of the path under test itself (tests/plan_decisions.py reads them off its saved
#    pre-activations), one boolean array per relu / leaky_relu call in call order, in the call's own layout.  The oracle then
#    differentiates the SAME piecewise-linear branch and the comparison is elementwise again.  Where a forced decision differs
#    from the oracle's own `v > 0` is counted in TIES['flips'], and the largest |v| among those inputs is kept in
#    TIES['flip_max']: the test asserts it is of the size of the tested arithmetic's round-off, i.e. that only genuinely
#    undecidable inputs were decided differently.
# eps = 0 and forced = None is the plain function.
TIES = {'eps': 0.0, 'side': 'on', 'count': 0, 'forced': None, 'pos': 0, 'flips': 0, 'flip_max': 0.0}


def _positive(v):
    forced = TIES.get('forced')
    if forced is not None:
        m = forced[TIES['pos']]
        TIES['pos'] += 1
        m = np.asarray(m.cpu() if hasattr(m, 'cpu') else m, dtype=bool)
        assert m.shape == v.shape, ('forced decision %d has shape %s, the call sees %s' % (TIES['pos'] - 1, m.shape, v.shape))
        dis = m != (v > 0)
        n = int(dis.sum())
        if n:
            TIES['flips'] += n
            TIES['flip_max'] = max(TIES['flip_max'], float(np.abs(v[dis]).max()))
        return m
    eps = TIES['eps']
    if eps <= 0:
        return v > 0
    TIES['count'] += int((np.abs(v) < eps).sum())
    return v > (-eps if TIES['side'] == 'on' else eps)


def relu(a):
    pos = _positive(a.v)
    out = Var(np.where(pos, a.v, 0), (a,))
    out.bw = lambda g: a.acc(g * pos)
    return out


def leaky_relu(a, slope):
    pos = _positive(a.v)
    out = Var(np.where(pos, a.v, a.v * slope), (a,))
    out.bw = lambda g: a.acc(g * np.where(pos, 1.0, slope))
    return out


def permute(a, axes):
    inv = np.argsort(axes)
    out = Var(np.transpose(a.v, axes), (a,))
    out.bw = lambda g: a.acc(np.transpose(g, inv))
    return out


def reshape(a, shape):
    out = Var(a.v.reshape(shape), (a,))
    out.bw = lambda g: a.acc(g.reshape(a.v.shape))
    return out


def getitem(a, idx):
    out = Var(a.v[idx], (a,))

    def bw(g):
        z = np.zeros_like(a.v)
        np.add.at(z, idx, g) if _fancy(idx) else z.__setitem__(idx, g)
        a.acc(z)
    out.bw = bw
    return out


def _fancy(idx):
    idx = idx if isinstance(idx, tuple) else (idx,)
    return any(isinstance(i, (np.ndarray, list)) for i in idx)


def cat(vs, axis):
    out = Var(np.concatenate([v.v for v in vs], axis=axis), tuple(vs))
    sizes = np.cumsum([0] + [v.v.shape[axis] for v in vs])

    def bw(g):
        for v, lo, hi in zip(vs, sizes[:-1], sizes[1:]):
            sl = [slice(None)] * g.ndim
            sl[axis] = slice(lo, hi)
            v.acc(g[tuple(sl)])
    out.bw = bw
    return out


def matmul(a, b):
    out = Var(np.matmul(a.v, b.v), (a, b))

    def bw(g):
        if a.needs:
            a.acc(_unbroadcast(np.matmul(g, np.swapaxes(b.v, -1, -2)), a.v.shape))
        if b.needs:
            b.acc(_unbroadcast(np.matmul(np.swapaxes(a.v, -1, -2), g), b.v.shape))
    out.bw = bw
    return out


def softmax(a, axis=-1):
    m = a.v.max(axis=axis, keepdims=True)
    e = np.exp(a.v - m)
    p = e / e.sum(axis=axis, keepdims=True)
    out = Var(p, (a,))
    out.bw = lambda g: a.acc(p * (g - (g * p).sum(axis=axis, keepdims=True)))
    return out


def masked_fill_from(e, mask, fill):
    """`adj = fill * ones; adj[mask] = e.view(-1)` (local_attention.py:40-41). mask: bool (C,J,J); e: (C,nnz)."""
    v = np.full(mask.shape, fill, dtype=e.v.dtype)
    v[mask] = e.v.reshape(-1)
    out = Var(v, (e,))
    out.bw = lambda g: e.acc(g[mask].reshape(e.v.shape))
    return out


# ---------------------------------------------------------------- NN primitives
def conv2d_k1(x, w, dilation=1, stride=1):
    """torch.nn.functional.conv2d for kernel (k,1), dilation (d,1), stride (s,1), no padding, no bias.
    x: (B,Cin,T,J)  w: (Cout,Cin,k,1)  ->  (B,Cout,Tout,J),  Tout = (T - d*(k-1) - 1)//s + 1."""
    B, Cin, T, J = x.v.shape
    Cout, _, k, _ = w.v.shape
    Tout = (T - dilation * (k - 1) - 1) // stride + 1
    taps = [x.v[:, :, tau * dilation: tau * dilation + (Tout - 1) * stride + 1: stride] for tau in range(k)]
    y = sum(np.einsum('bctj,oc->botj', taps[tau], w.v[:, :, tau, 0]) for tau in range(k))
    out = Var(y, (x, w))

    def bw(g):
        if w.needs:
            gw = np.stack([np.einsum('botj,bctj->oc', g, taps[tau]) for tau in range(k)], axis=2)[..., None]
            w.acc(gw)
        if x.needs:
            gx = np.zeros_like(x.v)
            for tau in range(k):
                gx[:, :, tau * dilation: tau * dilation + (Tout - 1) * stride + 1: stride] += \
                    np.einsum('botj,oc->bctj', g, w.v[:, :, tau, 0])
            x.acc(gx)
    out.bw = bw
    return out


def conv1d_1x1(x, w, b):
    """nn.Conv1d(kernel_size=1) with bias: x (N,Cin,L), w (Cout,Cin,1), b (Cout,) -> (N,Cout,L)."""
    y = np.einsum('ncl,oc->nol', x.v, w.v[:, :, 0]) + b.v[None, :, None]
    out = Var(y, (x, w, b))

    def bw(g):
        if w.needs:
            w.acc(np.einsum('nol,ncl->oc', g, x.v)[..., None])
        if b.needs:
            b.acc(g.sum(axis=(0, 2)))
        if x.needs:
            x.acc(np.einsum('nol,oc->ncl', g, w.v[:, :, 0]))
    out.bw = bw
    return out


def batch_norm2d(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d on (B,C,T,J). Train: batch mean / biased variance normalise, running stats updated in place
    with the unbiased variance.  Eval: running stats.  Returns the output Var."""
    ax = (0, 2, 3)
    sh = (1, -1, 1, 1)
    if training:
        n = x.v.shape[0] * x.v.shape[2] * x.v.shape[3]
        mean = x.v.mean(axis=ax)
        var = x.v.var(axis=ax)
        running_mean *= (1 - momentum)
        running_mean += momentum * mean
        running_var *= (1 - momentum)
        running_var += momentum * var * (n / max(n - 1, 1))
    else:
        mean, var = running_mean, running_var
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x.v - mean.reshape(sh)) * rstd.reshape(sh)
    out = Var(xhat * gamma.v.reshape(sh) + beta.v.reshape(sh), (x, gamma, beta))

    def bw(g):
        gamma.acc((g * xhat).sum(axis=ax))
        beta.acc(g.sum(axis=ax))
        if x.needs:
            gs = g * gamma.v.reshape(sh)
            if training:
                m1 = gs.mean(axis=ax).reshape(sh)
                m2 = (gs * xhat).mean(axis=ax).reshape(sh)
                x.acc((gs - m1 - xhat * m2) * rstd.reshape(sh))
            else:
                x.acc(gs * rstd.reshape(sh))
    out.bw = bw
    return out


def dropout_mask(a, keep_scaled):
    """y = a * keep_scaled, where keep_scaled is a constant array of 0 / 1/(1-p) entries."""
    return mul_const(a, keep_scaled)


def mpjpe(pred, target):
    """common/loss.py:5-11: mean over everything of the L2 norm over the last axis."""
    d = pred.v - target
    nrm = np.sqrt((d * d).sum(axis=-1))
    out = Var(np.asarray(nrm.mean()), (pred,))

    def bw(g):
        safe = np.where(nrm > 0, nrm, 1.0)
        pred.acc(g * d / safe[..., None] / nrm.size)
    out.bw = bw
    return out


def forced_oracle(run_oracle, decisions):
    """run_oracle() with every relu / leaky_relu decision taken from `decisions` (oracle/plan_decisions.py: what the path under test
    decided).  Returns (result, number of decisions that differ from the oracle's own, largest |input| among those)."""
    TIES.update(forced=list(decisions), pos=0, flips=0, flip_max=0.0)
    try:
        out = run_oracle()
        used, flips, fmax = TIES['pos'], TIES['flips'], TIES['flip_max']
    finally:
        TIES.update(forced=None, pos=0, flips=0, flip_max=0.0)
    assert used == len(decisions), ('the oracle made %d relu calls, %d decisions were supplied' % (used, len(decisions)))
    return out, flips, fmax
