"""Stand-alone forward passes of the reference's graph sub-modules on the `gast_hip` op set.

Inside `SpatioTemporalModel` these modules never run on their own: their arithmetic is part of the fused plan (gast_hip/engine.py).
The reference nevertheless exports them as ordinary `nn.Module`s (`from model.gast_net import *` leaks `LocalGraph`,
`MultiGlobalGraph`, `SingleGlobalGraph`; `model/sem_graph_conv.py` ships the channel-shared twin), so each gets a small plan of
its own over the same kernels:

    SemCHGraphConv / SemGraphConv   X.[W0|W1] GEMM -> masked adjacency softmax -> neighbour aggregation (+ bias)
                                    reference local_attention.py:35-53 / sem_graph_conv.py:35-52
    LocalGraph (both flavours)      G1 (4 C columns) -> AGG + bn_1/bn_2 statistics -> G2 + cat_bn -> ReLU (+ dropout)
                                    reference local_attention.py:130-151 / sem_graph_conv.py:130-153
    GlobalGraph                     [g | v_theta | v_phi] GEMM -> additive joint attention           global_attention.py:52-82
    MultiGlobalGraph                all heads in one GEMM -> attention -> cat_conv + cat_bn -> ReLU (+ dropout)   :103-130
    SingleGlobalGraph               one full-width head -> bn -> ReLU (+ dropout)                     :148-173
    GraphAttentionBlock             cat(x, local, global) as three K segments -> cat_conv + cat_bn -> ReLU      gast_net.py:22-33

Two paths per module.  Under torch.no_grad() (inference, feature extraction): the fused forward plans below (BatchNorm + ReLU applied
in the consumers' load prologues, statistics from the producers' epilogues).  When autograd has to record the call (a parameter or the
input requires a gradient): the same arithmetic composed from the four differentiable blocks of gast_hip/autograd_ops.py (Gemm, BnRelu,
SemchAgg, Attn), whose backward passes are the fused plan's own backward kernels -- the modules are trainable, including the gradient
with respect to their input, like the reference's.  fp32, device tensors, no CPU fallback.  Train mode uses batch statistics and
updates the running statistics exactly like nn.BatchNorm2d; nn.Dropout in train mode uses the library's counter-hash stream (same
distribution as torch's, not the same stream -- as in the fused plan).
"""
import functools

import torch

from gast_hip.engine import EPI_PLAIN, EPI_STATS, PRO_BNRELU, PRO_NONE, RowMap, ident

_OPS = None


def _on_input_device(fn):
    """kernels are enqueued on the current stream of the CURRENT device: make that the device of the input tensor"""
    @functools.wraps(fn)
    def wrapped(mod, x, *a, **k):
        if torch.is_tensor(x) and x.is_cuda:
            with torch.cuda.device(x.device):
                return fn(mod, x, *a, **k)
        return fn(mod, x, *a, **k)
    return wrapped


def _ops():
    global _OPS
    if _OPS is None:
        from gast_hip.binding import HipOps
        _OPS = HipOps()
    return _OPS


def _check(mod, x, ndim):
    if not x.is_cuda:
        raise RuntimeError('%s (MI355X build): input is on %s; this implementation has no CPU fallback' % (type(mod).__name__, x.device))
    if x.dim() != ndim:
        raise RuntimeError('%s: expected a %d-D input, got shape %s' % (type(mod).__name__, ndim, tuple(x.shape)))
    return x.contiguous().float()


def _needs_grad(mod, x):
    """does autograd have to record this call?  (then the differentiable composition runs instead of the fused forward plan)"""
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mod.parameters()))


def _pattern(mod, pat, dev, full=False):
    """device pattern table (+ nnz, max row degree[, max column degree]) of a 0/1 (J, J) pattern, cached on the module"""
    from model.local_attention import pattern_table
    cache = mod.__dict__.setdefault('_gast_pat', {})
    key = str(dev)
    if key not in cache:
        tab, nnz = pattern_table(pat)
        J = pat.shape[0]
        off = 2 + 2 * (J + 1) + 3 * nnz
        cache[key] = (tab.to(dev), nnz, int(tab[off]), int(tab[off + 1]))
    return cache[key] if full else cache[key][:3]


def _bn_state(ops, bn, partials, nblk, col0, n, count, scale, shift):
    """scale / shift of one BatchNorm2d for this call: batch statistics (+ running update) in train mode, running ones in eval"""
    dev = scale.device
    if bn.training:
        if not bn.track_running_stats or bn.momentum is None or not bn.affine:
            raise NotImplementedError('BatchNorm2d needs affine=True, track_running_stats=True and a numeric momentum')
        mean = torch.empty(n, dtype=torch.float32, device=dev)
        rstd = torch.empty(n, dtype=torch.float32, device=dev)
        ops.bn_finalize(partials, nblk, col0, n, count, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                        float(bn.momentum), float(bn.eps), scale, shift, mean, rstd)
    else:
        ops.bn_eval(bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), n, scale, shift)


def _dropout(mod_dropout, training, dev):
    """(use_drop, Dropout) of an nn.Dropout for this call"""
    if mod_dropout is None or not training or mod_dropout.p <= 0:
        return False, None
    from gast_hip.binding import Dropout, dropout_params
    thresh, inv_keep = dropout_params(float(mod_dropout.p))
    seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int32).to(dev)
    return True, Dropout(seed, thresh, inv_keep)


def _gemm_stats(ops, dom, N, segs, P, dev, bias=None):
    """GEMM + column statistics for the BatchNorm that follows: (out, partials, nblk)"""
    out = torch.empty(P, N, dtype=torch.float32, device=dev)
    nb = ops.gemm_row_blocks(P)
    part = torch.zeros(nb, N, 2, dtype=torch.float32, device=dev)
    ops.gemm(dom, N, segs, out, ident(dom[1]), epi=EPI_STATS, partials=part, bias=bias)
    return out, part, nb


# ------------------------------------------------------------------------------------------ differentiable compositions
def _drop_state(mod_dropout, training, dev):
    use, d = _dropout(mod_dropout, training, dev)
    return d if use else None


def _graph_conv_pair_grad(X, dom, convs, shared):
    """differentiable twin of _graph_conv_pair: Y (P, 2 Cout)"""
    from gast_hip.autograd_ops import Gemm, SemchAgg
    B, T, J = dom
    dev = X.device
    Cout = convs[0].out_features
    W4 = torch.cat([c.W[q].t() for c in convs for q in (0, 1)], dim=0)                       # [4 Cout][Cin]; autograd splits dW4 back
    H = Gemm.apply(X, W4, None, dom)
    metas, es = [], []
    for c in convs:
        pat = (c.adj if c.adj.dim() == 2 else c.adj[0]) > 0
        tab, nnz, dr, dc = _pattern(c, pat, dev, full=True)
        metas.append((tab, nnz, dr, dc))
        es.append(c.e.expand(Cout, nnz) if shared else c.e)                                    # (shared: autograd sums de over the channels)
    return SemchAgg.apply(H, es[0], es[1], (metas[0], metas[1], B * T, J, Cout))


def _bnrelu_grad(X, bn, drop=None, salt=0):
    from gast_hip.autograd_ops import BnRelu
    return BnRelu.apply(X, bn.weight, bn.bias, bn, drop, salt)


def _attention_grad(X, dom, heads):
    """differentiable twin of _attention: theta / phi folded into one C-vector per head by torch ops on the parameters"""
    from gast_hip.autograd_ops import Gemm, Attn
    B, T, J = dom
    C = X.shape[1]
    Wg, bg, va, ba, vc, bc = [], [], [], [], [], []
    for h in heads:
        Ci = h.inter_channels
        w = h.concat_project[0].weight.view(2 * Ci)
        Wg.append(h.g.weight.view(h.g_channels, C))
        bg.append(h.g.bias)
        va.append(h.theta.weight.view(Ci, C).t() @ w[:Ci])
        ba.append((w[:Ci] * h.theta.bias).sum().view(1))
        vc.append(h.phi.weight.view(Ci, C).t() @ w[Ci:])
        bc.append((w[Ci:] * h.phi.bias).sum().view(1))
    W = torch.cat(Wg + [torch.stack(va), torch.stack(vc)], dim=0)
    bias = torch.cat(bg + ba + bc)
    H = Gemm.apply(X, W, bias, dom)
    Cg = W.shape[0] - 2 * len(heads)
    Ck = torch.stack([h.C_k for h in heads])
    return Attn.apply(H[:, :Cg], H[:, Cg:], Ck, B * T, J, len(heads))


# ------------------------------------------------------------------------------------------ SemCH / Sem graph convolution
def _graph_conv_pair(ops, X, dom, convs, shared):
    """The two graph convolutions of a LocalGraph (or one, passed twice) on X (P, Cin): returns Y (P, 2 Cout) and the
    aggregation kernel's partial column sums.  convs = (conv_a, conv_b); shared: channel-shared adjacency (SemGraphConv: e is
    (1, nnz), broadcast over the channels) instead of the channel-wise one (SemCHGraphConv: e is (Cout, nnz))."""
    B, T, J = dom
    P, dev = X.shape[0], X.device
    Cout = convs[0].out_features
    W4 = torch.cat([c.W[q].t() for c in convs for q in (0, 1)], dim=0).contiguous()          # [4 Cout][Cin]
    H = torch.empty(P, 4 * Cout, dtype=torch.float32, device=dev)
    ops.gemm(dom, 4 * Cout, [dict(A=X, K=X.shape[1], map=ident(T), W=W4)], H, ident(T))
    As, tabs, degs = [], [], []
    for c in convs:
        pat = (c.adj if c.adj.dim() == 2 else c.adj[0]) > 0
        tab, nnz, dr = _pattern(c, pat, dev)
        e = c.e.expand(Cout, nnz).contiguous() if shared else c.e
        A = torch.empty(nnz + 1, Cout, dtype=torch.float32, device=dev)
        ops.semch_adj_fwd(e, tab, A)
        As.append(A); tabs.append(tab); degs.append(dr)
    F = B * T
    Y = torch.empty(P, 2 * Cout, dtype=torch.float32, device=dev)
    nba = ops.semch_agg_blocks(F, Cout)
    part = torch.empty(nba, 2 * Cout, 2, dtype=torch.float32, device=dev)
    # (the unrolled fixed-degree kernels exist for the (sym, con) degree pairs of the supported skeletons; anything else takes the CSR kernel)
    ops.semch_agg_fwd(H, F, J, Cout, As[0], tabs[0], As[1], tabs[1], Y, part, deg=(degs[0], degs[1]))
    return Y, part, nba


@_on_input_device
def graph_conv_forward(mod, x, shared):
    """SemCHGraphConv.forward (shared=False, reference local_attention.py:35-53) / SemGraphConv.forward (shared=True,
    sem_graph_conv.py:35-52): x (B, T, J, Cin) -> (B, T, J, Cout)"""
    x = _check(mod, x, 4)
    B, T, J, Cin = x.shape
    if _needs_grad(mod, x):
        Y = _graph_conv_pair_grad(x.reshape(B * T * J, Cin), (B, T, J), (mod, mod), shared)
    else:
        Y, _, _ = _graph_conv_pair(_ops(), x.view(B * T * J, Cin), (B, T, J), (mod, mod), shared)
    out = Y[:, :mod.out_features].reshape(B, T, J, mod.out_features)
    if mod.bias is not None:
        out = out + mod.bias.view(1, 1, -1)
    return out


@_on_input_device
def local_graph_forward(mod, x, shared=False, dropout2d=False):
    """LocalGraph.forward: x (B, T, J, C) -> (B, T, J, Cout)   (reference local_attention.py:130-151; sem_graph_conv.py:130-153
    for the channel-shared twin, whose dropout is nn.Dropout2d)"""
    x = _check(mod, x, 4)
    ops = _ops()
    B, T, J, C = x.shape
    P, dev, dom = B * T * J, x.device, (B, T, J)
    Co = mod.gcn_sym.out_features
    if _needs_grad(mod, x):
        from gast_hip.autograd_ops import Gemm
        Y = _graph_conv_pair_grad(x.reshape(P, C), dom, (mod.gcn_sym, mod.gcn_con), shared)
        if mod.gcn_sym.bias is not None or mod.gcn_con.bias is not None:
            zero = torch.zeros(Co, dtype=torch.float32, device=dev)
            Y = Y + torch.cat([zero if c.bias is None else c.bias for c in (mod.gcn_sym, mod.gcn_con)]).view(1, -1)
        Z = torch.cat([_bnrelu_grad(Y[:, :Co], mod.bn_1), _bnrelu_grad(Y[:, Co:], mod.bn_2)], dim=1)
        L = Gemm.apply(Z, mod.cat_conv.weight.view(Co, 2 * Co), None, dom)
        if dropout2d:
            out = _bnrelu_grad(L, mod.cat_bn).view(B, T, J, Co)
            if mod.dropout is not None and mod.training:
                out = mod.dropout(out.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
            return out
        return _bnrelu_grad(L, mod.cat_bn, _drop_state(mod.dropout, mod.training, dev), 1).view(B, T, J, Co)
    Y, partY, nba = _graph_conv_pair(ops, x.view(P, C), dom, (mod.gcn_sym, mod.gcn_con), shared)
    if mod.gcn_sym.bias is not None or mod.gcn_con.bias is not None:
        # biased convolutions (SemGraphConv default): the bias shifts the BatchNorm input; statistics from a column pass
        for half, c in enumerate((mod.gcn_sym, mod.gcn_con)):
            if c.bias is not None:
                Y[:, half * Co:(half + 1) * Co] += c.bias.view(1, -1)
        nba = ops.rowwise_blocks(P, 2 * Co)
        partY = torch.stack([Y.sum(0), (Y * Y).sum(0)], dim=-1).unsqueeze(0).contiguous()
        nba = 1
    scY = torch.empty(2 * Co, dtype=torch.float32, device=dev)
    shY = torch.empty(2 * Co, dtype=torch.float32, device=dev)
    _bn_state(ops, mod.bn_1, partY, nba, 0, Co, P, scY[:Co], shY[:Co])
    _bn_state(ops, mod.bn_2, partY, nba, Co, Co, P, scY[Co:], shY[Co:])
    Wlc = mod.cat_conv.weight.view(Co, 2 * Co)
    L, partL, nb = _gemm_stats(ops, dom, Co, [dict(A=Y, K=2 * Co, map=ident(T), W=Wlc, pro=PRO_BNRELU, scale=scY, shift=shY)], P, dev)
    scL = torch.empty(Co, dtype=torch.float32, device=dev)
    shL = torch.empty(Co, dtype=torch.float32, device=dev)
    _bn_state(ops, mod.cat_bn, partL, nb, 0, Co, P, scL, shL)
    out = torch.empty(P, Co, dtype=torch.float32, device=dev)
    if dropout2d:
        ops.bnrelu_apply(L, P, Co, scL, shL, out)
        out = out.view(B, T, J, Co)
        if mod.dropout is not None and mod.training:
            out = mod.dropout(out.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)      # nn.Dropout2d: whole (b, c) planes, torch's own stream
        return out
    use_drop, drop = _dropout(mod.dropout, mod.training, dev)
    ops.bnrelu_apply(L, P, Co, scL, shL, out, use_drop=use_drop, salt=1, drop=drop)
    return out.view(B, T, J, Co)


# ------------------------------------------------------------------------------------------ global attention
def _heads_gemm(ops, X, dom, heads):
    """[g (all heads) | a (all heads) | c (all heads)] = X . W^T + bias with theta / phi folded into one C-vector per head
    (f_ij = w_theta.theta_i + w_phi.phi_j is rank-1; reference global_attention.py:60-74)"""
    T = dom[1]
    C = X.shape[1]
    Wg, bg, va, ba, vc, bc = [], [], [], [], [], []
    for h in heads:
        Ci = h.inter_channels
        w = h.concat_project[0].weight.view(2 * Ci)
        Wg.append(h.g.weight.view(h.g_channels, C))
        bg.append(h.g.bias)
        va.append(h.theta.weight.view(Ci, C).t() @ w[:Ci])
        ba.append((w[:Ci] * h.theta.bias).sum().view(1))
        vc.append(h.phi.weight.view(Ci, C).t() @ w[Ci:])
        bc.append((w[Ci:] * h.phi.bias).sum().view(1))
    W = torch.cat(Wg + [torch.stack(va), torch.stack(vc)], dim=0).contiguous()
    bias = torch.cat(bg + ba + bc).contiguous()
    N = W.shape[0]
    H = torch.empty(X.shape[0], N, dtype=torch.float32, device=X.device)
    ops.gemm(dom, N, [dict(A=X, K=C, map=ident(T), W=W)], H, ident(T), bias=bias)
    return H, N - 2 * len(heads)


def _attention(ops, X, dom, heads):
    """additive joint attention of `heads` on X (P, C): Y (P, sum of g widths)"""
    B, T, J = dom
    H, Cg = _heads_gemm(ops, X, dom, heads)
    Ck = torch.stack([h.C_k for h in heads]).contiguous()
    Ya = torch.empty(X.shape[0], Cg, dtype=torch.float32, device=X.device)
    ops.attn_fwd(H[:, :Cg], H[:, Cg:], Ck, B * T, J, Cg, len(heads), Ya)
    return Ya


@_on_input_device
def global_graph_forward(mod, x):
    """GlobalGraph.forward: x (B*T, C, J) -> (B*T, g_channels, J)   (reference global_attention.py:52-82)"""
    x = _check(mod, x, 3)
    F, C, J = x.shape
    X = x.permute(0, 2, 1).contiguous().view(F * J, C)
    Ya = _attention_grad(X, (F, 1, J), [mod]) if _needs_grad(mod, x) else _attention(_ops(), X, (F, 1, J), [mod])
    return Ya.view(F, J, mod.g_channels).permute(0, 2, 1).contiguous()


@_on_input_device
def multi_global_forward(mod, x):
    """MultiGlobalGraph.forward: x (B, T, J, C) -> (B, T, J, C)   (reference global_attention.py:103-130)"""
    x = _check(mod, x, 4)
    ops = _ops()
    B, T, J, C = x.shape
    P, dev, dom = B * T * J, x.device, (B, T, J)
    if _needs_grad(mod, x):
        from gast_hip.autograd_ops import Gemm
        Ya = _attention_grad(x.reshape(P, C), dom, list(mod.attentions))
        G = Gemm.apply(Ya, mod.cat_conv.weight.view(C, C), None, dom)
        return _bnrelu_grad(G, mod.cat_bn, _drop_state(mod.dropout, mod.training, dev), 2).view(B, T, J, C)
    Ya = _attention(ops, x.view(P, C), dom, list(mod.attentions))
    Wgc = mod.cat_conv.weight.view(C, C)
    G, partG, nb = _gemm_stats(ops, dom, C, [dict(A=Ya, K=C, map=ident(T), W=Wgc)], P, dev)
    sc = torch.empty(C, dtype=torch.float32, device=dev)
    sh = torch.empty(C, dtype=torch.float32, device=dev)
    _bn_state(ops, mod.cat_bn, partG, nb, 0, C, P, sc, sh)
    out = torch.empty(P, C, dtype=torch.float32, device=dev)
    use_drop, drop = _dropout(mod.dropout, mod.training, dev)
    ops.bnrelu_apply(G, P, C, sc, sh, out, use_drop=use_drop, salt=2, drop=drop)
    return out.view(B, T, J, C)


@_on_input_device
def single_global_forward(mod, x):
    """SingleGlobalGraph.forward: x (B, T, J, C) -> (B, T, J, C); one head whose value projection keeps the full width
    (output_channels == in_channels, the only configuration the reference's bn(in_channels) accepts; global_attention.py:148-173)"""
    x = _check(mod, x, 4)
    ops = _ops()
    B, T, J, C = x.shape
    if mod.attentions.g_channels != C:
        raise RuntimeError('SingleGlobalGraph: the head emits %d channels but bn expects %d (the reference fails the same way unless '
                           'output_channels == in_channels)' % (mod.attentions.g_channels, C))
    P, dev = B * T * J, x.device
    if _needs_grad(mod, x):
        Ya = _attention_grad(x.reshape(P, C), (B, T, J), [mod.attentions])
        return _bnrelu_grad(Ya, mod.bn, _drop_state(mod.dropout, mod.training, dev), 3).view(B, T, J, C)
    Ya = _attention(ops, x.view(P, C), (B, T, J), [mod.attentions])
    nb = ops.rowwise_blocks(P, C)
    # statistics of the attention output (no GEMM in between): one column pass
    part = torch.stack([Ya.sum(0), (Ya * Ya).sum(0)], dim=-1).unsqueeze(0).contiguous()
    sc = torch.empty(C, dtype=torch.float32, device=dev)
    sh = torch.empty(C, dtype=torch.float32, device=dev)
    _bn_state(ops, mod.bn, part, 1, 0, C, P, sc, sh)
    out = torch.empty(P, C, dtype=torch.float32, device=dev)
    use_drop, drop = _dropout(mod.dropout, mod.training, dev)
    ops.bnrelu_apply(Ya, P, C, sc, sh, out, use_drop=use_drop, salt=3, drop=drop)
    return out.view(B, T, J, C)


# ------------------------------------------------------------------------------------------ GraphAttentionBlock
@_on_input_device
def graph_attention_block_forward(mod, x):
    """GraphAttentionBlock.forward: x (B, C, T, J) -> (B, 2 C_out, T, J)   (reference gast_net.py:22-33)"""
    x = _check(mod, x, 4)
    ops = _ops()
    B, C, T, J = x.shape
    xr = x.permute(0, 2, 3, 1).contiguous()
    P, dev, dom = B * T * J, x.device, (B, T, J)
    L = mod.local_graph_layer(xr).reshape(P, -1)
    G = mod.global_graph_layer(xr).reshape(P, -1)
    Co2 = mod.cat_conv.weight.shape[0]
    Wbc = mod.cat_conv.weight.view(Co2, -1)
    if _needs_grad(mod, x):
        from gast_hip.autograd_ops import Gemm
        O = Gemm.apply(torch.cat([xr.reshape(P, C), L, G], dim=1), Wbc, None, dom)
        return _bnrelu_grad(O, mod.cat_bn).view(B, T, J, Co2).permute(0, 3, 1, 2).contiguous()
    kL, kG = L.shape[1], G.shape[1]
    segs = [dict(A=xr.view(P, C), K=C, map=ident(T), W=Wbc[:, :C]), dict(A=L, K=kL, map=ident(T), W=Wbc[:, C:C + kL]),
            dict(A=G, K=kG, map=ident(T), W=Wbc[:, C + kL:C + kL + kG])]
    O, partO, nb = _gemm_stats(ops, dom, Co2, segs, P, dev)
    sc = torch.empty(Co2, dtype=torch.float32, device=dev)
    sh = torch.empty(Co2, dtype=torch.float32, device=dev)
    _bn_state(ops, mod.cat_bn, partO, nb, 0, Co2, P, sc, sh)
    out = torch.empty(P, Co2, dtype=torch.float32, device=dev)
    ops.bnrelu_apply(O, P, Co2, sc, sh, out)
    return out.view(B, T, J, Co2).permute(0, 3, 1, 2).contiguous()
