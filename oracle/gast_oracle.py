"""CPU restatement (numpy) of the reference's spatio-temporal hot path -- TEST INFRASTRUCTURE ONLY.

This is the parity ORACLE for the HIP path.  It restates, function by function, what
`/root/reference/model/{gast_net,local_attention,global_attention,sem_graph_conv}.py` compute, in the reference's own
tensor layout `(B, C, T, J)`, on top of the tiny numpy autodiff in `np_autograd.py` (so gradients come from
per-primitive VJPs, not from a second hand-derivation of the model).  Every function cites the reference lines it
follows.  It is pinned against fixtures produced by running the reference itself (tests/golden/make_golden.py ->
tests/golden/*.npz; checked by tests/test_oracle_golden.py).  The reference has no golden vectors of its own
(SURVEY.md section 8c).

The primitives come from a backend module: `np_autograd` (default: numpy, float64, the pinned oracle) or, inside
`with use_backend(torch_ops):`, stock PyTorch operators on any device -- the same restatement then is the "stock
PyTorch-ROCm" comparator of SURVEY.md section 8(d) and a GPU-side second reference (see oracle/torch_ops.py).

Only tests/, `__graft_entry__.smoke()` and `bench.py`'s baseline legs import this module.  The product never does.
"""
import contextlib

import numpy as np

from . import np_autograd as ag

NEG_FILL = -9e15  # local_attention.py:40


@contextlib.contextmanager
def use_backend(mod):
    """Run the restatement on another primitive set (oracle.torch_ops) for the duration of the block."""
    global ag
    old, ag = ag, mod
    try:
        yield
    finally:
        ag = old


# ------------------------------------------------------------------------------------------------ skeleton / patterns
def adj_from_parents(parents, dtype=np.float32):
    """common/graph_utils.py:27-45 (`adj_mx_from_skeleton` -> `adj_mx_from_edges`, dense): symmetric 0/1 adjacency
    of the kinematic tree plus self loops, row-normalised."""
    J = len(parents)
    a = np.zeros((J, J), dtype=np.float64)
    for i, p in enumerate(parents):
        if p >= 0:
            a[i, p] = 1.0
            a[p, i] = 1.0
    a = a + np.eye(J)
    a = a / a.sum(axis=1, keepdims=True)
    return a.astype(dtype)


JOINT_GROUPS = {  # local_attention.py:66-87
    17: dict(distal=[3, 6, 10, 13, 16], left=[4, 5, 6, 11, 12, 13], right=[1, 2, 3, 14, 15, 16]),
    16: dict(distal=[3, 6, 9, 12, 15], left=[4, 5, 6, 10, 11, 12], right=[1, 2, 3, 13, 14, 15]),
    15: dict(distal=[4, 7, 10, 13], left=[2, 3, 4, 8, 9, 10], right=[5, 6, 7, 11, 12, 13]),
    19: dict(distal=[3, 4, 7, 8, 12, 15, 18], left=[5, 6, 7, 8, 13, 14, 15], right=[1, 2, 3, 4, 16, 17, 18]),
}


def local_graph_adjacencies(adj):
    """local_attention.py:66-114: the symmetry graph and the 1st/2nd-order connection graph.
    Returns (adj_sym, adj_con) as float arrays; only their `> 0` pattern reaches the arithmetic (:24)."""
    adj = np.asarray(adj, dtype=np.float64)
    J = adj.shape[0]
    if J not in JOINT_GROUPS:
        raise KeyError("The dimension of adj matrix is wrong!")  # local_attention.py:89-90
    grp = JOINT_GROUPS[J]
    left, right, distal = grp['left'], grp['right'], grp['distal']
    adj_sym = np.zeros_like(adj)
    for i in range(J):
        adj_sym[i, i] = 1
        if i in left:
            adj_sym[i, right[left.index(i)]] = 1.0
        if i in right:
            adj_sym[i, left[right.index(i)]] = 1.0
    a1 = adj.copy()
    a1[distal] = 0
    a2 = adj @ adj
    keep = np.zeros(J, dtype=bool)
    keep[distal] = True
    a2[~keep] = 0
    return adj_sym, a1 + a2


# ------------------------------------------------------------------------------------------------ model pieces
class _P:
    """Parameter/buffer store: wraps a name->ndarray state dict into tape leaves (parameters) and raw buffers."""

    def __init__(self, state, dtype):
        self.dtype = dtype
        self.leaves = {}
        self.buf = {}
        for k, v in state.items():
            if k.endswith('running_mean') or k.endswith('running_var'):
                self.buf[k] = ag.asarray(v, dtype)
            elif k.endswith('num_batches_tracked'):
                self.buf[k] = ag.asarray(v, None)
            else:
                self.leaves[k] = ag.leaf(ag.asarray(v, dtype))

    def p(self, name):
        return self.leaves[name]


def _bn(P, prefix, x, training):
    y = ag.batch_norm2d(x, P.p(prefix + '.weight'), P.p(prefix + '.bias'),
                        P.buf[prefix + '.running_mean'], P.buf[prefix + '.running_var'], training)
    if training:
        P.buf[prefix + '.num_batches_tracked'] = P.buf[prefix + '.num_batches_tracked'] + 1
    return y


def _dropout(x, p, rng):
    if p <= 0:
        return x
    if hasattr(ag, 'dropout'):       # stock backend: its own RNG stream
        return ag.dropout(x, p)
    if rng is None:
        return x
    keep = (rng.random(x.v.shape) >= p).astype(x.v.dtype) / (1.0 - p)
    return ag.dropout_mask(x, keep)


def sem_ch_graph_conv(P, prefix, x, adj_pattern):
    """SemCHGraphConv.forward, local_attention.py:35-53.  x: (B,T,J,C) -> (B,T,J,Cout)."""
    W = P.p(prefix + '.W')
    e = P.p(prefix + '.e')
    Cout = W.v.shape[2]
    J = adj_pattern.shape[0]
    h0 = ag.matmul(x, ag.getitem(W, 0))                                   # :37  (B,T,J,C)
    h1 = ag.matmul(x, ag.getitem(W, 1))                                   # :38
    h0 = ag.permute(ag.reshape(h0, h0.shape + (1,)), (0, 1, 3, 2, 4))     # unsqueeze/transpose -> (B,T,C,J,1)
    h1 = ag.permute(ag.reshape(h1, h1.shape + (1,)), (0, 1, 3, 2, 4))
    m = np.broadcast_to(adj_pattern > 0, (Cout, J, J))                    # :23-24
    adj = ag.softmax(ag.masked_fill_from(e, m, NEG_FILL), axis=2)         # :40-42
    E = ag.eye(J, x)                                                      # :44-45
    out = ag.add(ag.matmul(ag.mul_const(adj, E), h0), ag.matmul(ag.mul_const(adj, 1 - E), h1))  # :47
    out = ag.reshape(ag.permute(out, (0, 1, 3, 2, 4)), out.shape[:2] + (J, Cout))               # :48
    return out


def local_graph(P, prefix, x, adj, training, p_drop, rng):
    """LocalGraph.forward, local_attention.py:130-151.  x: (B,T,J,C) -> (B,T,J,C)."""
    adj_sym, adj_con = local_graph_adjacencies(adj)
    xs = sem_ch_graph_conv(P, prefix + '.gcn_sym', x, adj_sym)            # :132
    ys = sem_ch_graph_conv(P, prefix + '.gcn_con', x, adj_con)            # :133
    xs = ag.permute(xs, (0, 3, 1, 2))                                     # :136
    ys = ag.permute(ys, (0, 3, 1, 2))
    xs = ag.relu(_bn(P, prefix + '.bn_1', xs, training))                  # :139
    ys = ag.relu(_bn(P, prefix + '.bn_2', ys, training))
    out = ag.cat([xs, ys], axis=1)                                        # :142
    out = _bn(P, prefix + '.cat_bn', ag.conv2d_k1(out, P.p(prefix + '.cat_conv.weight')), training)  # :143
    out = _dropout(ag.relu(out), p_drop, rng)                             # :145-148
    return ag.permute(out, (0, 2, 3, 1))                                  # :149


def global_graph_head(P, prefix, x):
    """GlobalGraph.forward, global_attention.py:52-82.  x: (BT, C, N) -> (BT, Cg, N)."""
    g_x = ag.conv1d_1x1(x, P.p(prefix + '.g.weight'), P.p(prefix + '.g.bias'))            # :56
    g_x = ag.permute(g_x, (0, 2, 1))                                                      # :57 (BT,N,Cg)
    theta = ag.conv1d_1x1(x, P.p(prefix + '.theta.weight'), P.p(prefix + '.theta.bias'))  # :60 (BT,Ci,N)
    phi = ag.conv1d_1x1(x, P.p(prefix + '.phi.weight'), P.p(prefix + '.phi.bias'))        # :62
    Ci = theta.shape[1]
    w = ag.reshape(P.p(prefix + '.concat_project.0.weight'), (1, 2 * Ci))                 # Conv2d(2Ci->1,1x1) :40-42
    # :66-72: f[b,i,j] = sum_c w[c] theta[b,c,i] + sum_c w[Ci+c] phi[b,c,j]  (expand + cat + 1x1 conv, restated)
    a = ag.matmul(ag.getitem(w, (slice(None), slice(0, Ci))), theta)                      # (BT,1,N)
    c = ag.matmul(ag.getitem(w, (slice(None), slice(Ci, 2 * Ci))), phi)                   # (BT,1,N)
    f = ag.add(ag.permute(a, (0, 2, 1)), c)                                               # (BT,N,N): a_i + c_j
    att = ag.leaky_relu(f, 0.2)                                                           # :74
    att = ag.add(ag.softmax(att, axis=-1), P.p(prefix + '.C_k'))                          # :76
    y = ag.matmul(att, g_x)                                                               # :78 (BT,N,Cg)
    return ag.permute(y, (0, 2, 1))                                                       # :79-80


def multi_global_graph(P, prefix, x, training, p_drop, rng):
    """MultiGlobalGraph.forward, global_attention.py:103-130.  x: (B,T,J,C) -> (B,T,J,C)."""
    B, T, J, C = x.shape
    n_heads = len([k for k in P.leaves if k.startswith(prefix + '.attentions.') and k.endswith('.C_k')])
    xx = ag.permute(ag.reshape(x, (B * T, J, C)), (0, 2, 1))                              # :106-109
    heads = [global_graph_head(P, '%s.attentions.%d' % (prefix, h), xx) for h in range(n_heads)]
    xx = ag.cat(heads, axis=1)                                                            # :111
    xx = ag.reshape(ag.permute(xx, (0, 2, 1)), (B, T, J, C))                              # :114-118
    xx = ag.permute(xx, (0, 3, 1, 2))                                                     # :121
    xx = ag.relu(_bn(P, prefix + '.cat_bn', ag.conv2d_k1(xx, P.p(prefix + '.cat_conv.weight')), training))  # :122
    xx = _dropout(xx, p_drop, rng)                                                        # :124-125
    return ag.permute(xx, (0, 2, 3, 1))                                                   # :128


def graph_attention_block(P, prefix, x, adj, training, p_drop, rng):
    """GraphAttentionBlock.forward, gast_net.py:22-33.  x: (B,C,T,J) -> (B,2C,T,J)."""
    x = ag.permute(x, (0, 2, 3, 1))                                                       # :24
    residual = x
    x_ = local_graph(P, prefix + '.local_graph_layer', x, adj, training, p_drop, rng)     # :26
    y_ = multi_global_graph(P, prefix + '.global_graph_layer', x, training, p_drop, rng)  # :27
    x = ag.cat([residual, x_, y_], axis=-1)                                               # :28
    x = ag.permute(x, (0, 3, 1, 2))                                                       # :31
    return ag.relu(_bn(P, prefix + '.cat_bn', ag.conv2d_k1(x, P.p(prefix + '.cat_conv.weight')), training))  # :32


class OracleModel:
    """Restatement of SpatioTemporalModel (`variant='dilated'`, gast_net.py:107-177; `variant='dense'`: its `dense=True`
    ablation, :145-146 -- temporal kernels of width 2*pad+1 with dilation 1) and SpatioTemporalModelOptimized1f
    (`variant='strided'`, gast_net.py:180-251)."""

    def __init__(self, adj, filter_widths, channels, causal=False, dropout=0.0, variant='dilated', dtype=np.float64):
        for fw in filter_widths:
            assert fw % 2 != 0, 'Only odd filter widths are supported'   # gast_net.py:46-47
        self.adj = np.asarray(adj)
        self.fw = list(filter_widths)
        self.channels = channels
        self.causal = causal
        self.dropout = dropout
        self.variant = variant
        self.dtype = dtype
        # gast_net.py:57,139-143 / :215-220
        self.pad = [self.fw[0] // 2]
        self.causal_shift = [self.fw[0] // 2 if causal else 0]
        nd = self.fw[0]
        self.dil = [1]
        for i in range(1, len(self.fw)):
            self.pad.append((self.fw[i] - 1) * nd // 2)
            if variant != 'strided':
                self.causal_shift.append((self.fw[i] // 2 * nd) if causal else 0)
            else:
                self.causal_shift.append((self.fw[i] // 2) if causal else 0)
            self.dil.append(nd)
            nd *= self.fw[i]

    def receptive_field(self):
        return 1 + 2 * sum(self.pad)                                     # gast_net.py:62-69

    def forward(self, state, x, training=False, rng=None, P=None):
        """state: name->ndarray (reference state_dict layout); x: (B,T,J,2).  Returns (y Var (B,T',J,3), P).
        With training=True the BatchNorm buffers inside P.buf are updated as the reference would.  P: reuse a parameter store
        built earlier (`_P(state, dtype)`) instead of copying `state` again -- a training loop over persistent leaves."""
        if P is None:
            P = _P(state, self.dtype)
        p_drop = self.dropout if training else 0.0
        xv = ag.const(ag.asarray(x, self.dtype))
        assert xv.v.ndim == 4                                                              # gast_net.py:93-95
        h = ag.permute(xv, (0, 3, 1, 2))                                                   # :162
        h = _bn(P, 'init_bn', h, training)                                                 # :163
        strided = self.variant == 'strided'
        h = ag.conv2d_k1(h, P.p('expand_conv.weight'), stride=self.fw[0] if strided else 1)
        h = ag.relu(_bn(P, 'expand_bn', h, training))                                      # :164
        h = graph_attention_block(P, 'layers_graph_conv.0', h, self.adj, training, p_drop, rng)  # :165
        for i in range(len(self.pad) - 1):                                                 # :167
            k = self.fw[i + 1]
            if strided:
                start = self.causal_shift[i + 1] + k // 2                                  # :243
                res = ag.getitem(h, (slice(None), slice(None), slice(start, None, k)))
                c = ag.conv2d_k1(h, P.p('layers_conv.%d.weight' % (2 * i)), stride=k)      # :222,246
            else:
                pad, shift = self.pad[i + 1], self.causal_shift[i + 1]
                res = ag.getitem(h, (slice(None), slice(None), slice(pad + shift, h.shape[2] - pad + shift)))  # :170
                dilation = 1 if self.variant == 'dense' else self.dil[i + 1]                   # :145-146
                c = ag.conv2d_k1(h, P.p('layers_conv.%d.weight' % (2 * i)), dilation=dilation)  # :173
            c = ag.relu(_bn(P, 'layers_bn.%d' % (2 * i), c, training))
            c = ag.conv2d_k1(c, P.p('layers_conv.%d.weight' % (2 * i + 1)))                # :174
            c = _dropout(ag.relu(_bn(P, 'layers_bn.%d' % (2 * i + 1), c, training)), p_drop, rng)
            h = ag.add(res, c)
            h = graph_attention_block(P, 'layers_graph_conv.%d' % (i + 1), h, self.adj, training, p_drop, rng)  # :176
        y = ag.conv2d_k1(h, P.p('shrink.weight'))                                          # :99
        y = ag.permute(y, (0, 2, 3, 1))                                                    # :102
        return y, P

    def loss_and_grads(self, state, x, y3d, training=True, rng=None):
        """mpjpe(model(x), y3d) (main.py:230-237) and its gradient w.r.t. every parameter."""
        y, P = self.forward(state, x, training=training, rng=rng)
        loss = ag.mpjpe(y, ag.asarray(y3d, self.dtype))
        ag.backward(loss)
        grads = {k: (v.g if v.g is not None else v.v * 0) for k, v in P.leaves.items()}
        lv = loss.v.detach() if hasattr(loss.v, 'detach') else loss.v
        return float(lv), y.v, grads, P.buf

    def output_grads(self, state, x, dy, training=True):
        """Gradients of sum(y * dy) for an arbitrary upstream gradient dy (used for at-size GPU parity)."""
        y, P = self.forward(state, x, training=training)
        ag.backward(y, seed=ag.asarray(dy, self.dtype))
        grads = {k: (v.g if v.g is not None else v.v * 0) for k, v in P.leaves.items()}
        return y.v, grads, P.buf


def sem_graph_conv(x, W, e, adj_pattern, bias=None):
    """SemGraphConv.forward (dead twin), sem_graph_conv.py:35-52: ONE (J,J) masked-softmax adjacency shared by all
    channels.  x: (B,T,J,Cin) ndarray -> (B,T,J,Cout) ndarray (forward only; used for the f3 row)."""
    J = adj_pattern.shape[0]
    h0 = x @ W[0]
    h1 = x @ W[1]
    a = np.full((J, J), NEG_FILL, dtype=x.dtype)
    a[adj_pattern > 0] = e.reshape(-1)
    a = a - a.max(axis=1, keepdims=True)
    a = np.exp(a)
    a = a / a.sum(axis=1, keepdims=True)
    E = np.eye(J, dtype=x.dtype)
    out = np.matmul(a * E, h0) + np.matmul(a * (1 - E), h1)
    if bias is not None:
        out = out + bias.reshape(1, 1, -1)
    return out
