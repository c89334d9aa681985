"""Differentiable building blocks of the STAND-ALONE graph sub-modules (gast_hip/modules.py) on the `gast_hip` op set.

Inside `SpatioTemporalModel` forward and backward are one fused plan (gast_hip/engine.py).  The reference also exports its
sub-modules -- `SemCHGraphConv`, `LocalGraph`, `GlobalGraph`, `MultiGlobalGraph`, `SingleGlobalGraph`, `GraphAttentionBlock`, the
channel-shared `SemGraphConv` twin -- as ordinary trainable `nn.Module`s (reference local_attention.py:35-53,130-151,
global_attention.py:52-82,103-130,148-173, gast_net.py:22-33, sem_graph_conv.py:35-52,130-153).  Their stand-alone training path is
built from four `torch.autograd.Function`s, each a forward kernel paired with the backward kernels the fused plan uses:

    Gemm      Y = X . W^T + b             gast_gemm            | gast_gemm (dX = dY . W), gast_wgrad (dW = dY^T . X), column sum
    BnRelu    Z = drop(relu(bn(X)))       gast_bn_finalize /   | gast_bnrelu_bwd_mask -> gast_bn_bwd_finalize_multi -> gast_bn_bwd_apply
                                          _eval, _bnrelu_apply |
    SemchAgg  masked adjacency softmax +  gast_semch_adj_fwd,  | gast_semch_agg_bwd (dH, dA), gast_semch_adj_bwd (de)
              neighbour aggregation       gast_semch_agg_fwd   |
    Attn      additive joint attention    gast_attn_fwd        | gast_attn_bwd (dG, da / dc, dC_k)

Everything parameter-sized around them (stacking W0|W1, folding theta / phi into one vector per head, broadcasting the shared
adjacency) is plain torch on the parameters, so autograd unfolds those gradients by itself.  fp32 device tensors, no CPU fallback.
Gradients w.r.t. the module INPUT are produced as well (the reference's modules are differentiable in x)."""
import torch

from gast_hip.engine import BNState, ident

_OPS = None


def _ops():
    global _OPS
    if _OPS is None:
        from gast_hip.binding import HipOps
        _OPS = HipOps()
    return _OPS


def _pad4(t, dim):
    """zero-pad dimension `dim` of a 2-D tensor to a multiple of 4 (16-byte rows / K segments of the fp32 kernels)"""
    n = t.shape[dim]
    n4 = (n + 3) // 4 * 4
    if n4 == n:
        return t.contiguous()
    shape = list(t.shape)
    shape[dim] = n4
    out = torch.zeros(shape, dtype=t.dtype, device=t.device)
    out.narrow(dim, 0, n).copy_(t)
    return out


class Gemm(torch.autograd.Function):
    """Y (P, N) = X (P, K) . W (N, K)^T + bias; dom = (B, T, J) with P = B*T*J (the row domain of the kernels)."""

    @staticmethod
    def forward(ctx, X, W, bias, dom):
        ops = _ops()
        P, K = X.shape
        N = W.shape[0]
        Xp, Wp = _pad4(X.float(), 1), _pad4(W.float(), 1)
        Y = torch.empty(P, N, dtype=torch.float32, device=X.device)
        ops.gemm(dom, N, [dict(A=Xp, K=Xp.shape[1], map=ident(dom[1]), W=Wp)], Y, ident(dom[1]),
                 bias=None if bias is None else bias.float().contiguous())
        ctx.save_for_backward(Xp, Wp)
        ctx.dom, ctx.K, ctx.N, ctx.has_bias = dom, K, N, bias is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        ops = _ops()
        Xp, Wp = ctx.saved_tensors
        dom, K, N = ctx.dom, ctx.K, ctx.N
        T = dom[1]
        P = Xp.shape[0]
        dYp = _pad4(dY.float(), 1)                       # (P, N4): the K operand of the input gradient / the P operand of dW
        N4, K4 = dYp.shape[1], Xp.shape[1]
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            WT = torch.zeros(K4, N4, dtype=torch.float32, device=dY.device)      # [K][N]: dX = dY . W
            WT[:, :N] = Wp.t()
            dXp = torch.empty(P, K4, dtype=torch.float32, device=dY.device)
            ops.gemm(dom, K4, [dict(A=dYp, K=N4, map=ident(T), W=WT)], dXp, ident(T))
            dX = dXp[:, :K]
        if ctx.needs_input_grad[1]:
            dWp = torch.zeros(N4, K4, dtype=torch.float32, device=dY.device)     # split-M partial tiles are accumulated with atomics
            ops.wgrad(dom, dYp, N4, ident(T), [dict(Q=Xp, S=K4, map=ident(T), wcol0=0)], dWp, zero_first=False)
            dW = dWp[:N, :K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dY.sum(dim=0)
        return dX, dW, db, None


class BnRelu(torch.autograd.Function):
    """Z = dropout(relu(batchnorm(X))) over the columns of X (P, N) -- X may be a column slice of a wider tensor.  Train mode: batch
    statistics (+ the running-statistic update of nn.BatchNorm2d); eval mode: running statistics, whose backward is the fixed affine
    map's (the batch terms vanish: count -> infinity, as in the fused plan)."""

    @staticmethod
    def forward(ctx, X, weight, bias, bn, drop, salt):
        ops = _ops()
        P, N = X.shape
        dev = X.device
        scale = torch.empty(N, dtype=torch.float32, device=dev)
        shift = torch.empty(N, dtype=torch.float32, device=dev)
        mean = torch.empty(N, dtype=torch.float32, device=dev)
        rstd = torch.empty(N, dtype=torch.float32, device=dev)
        if bn.training:
            if not bn.track_running_stats or bn.momentum is None or not bn.affine:
                raise NotImplementedError('BatchNorm2d needs affine=True, track_running_stats=True and a numeric momentum')
            xd = X.double()
            part = torch.stack([xd.sum(0), (xd * xd).sum(0)], dim=-1).float().unsqueeze(0).contiguous()       # one statistics block
            ops.bn_finalize(part, 1, 0, N, P, weight, bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                            float(bn.momentum), float(bn.eps), scale, shift, mean, rstd)
            count = P
        else:
            ops.bn_eval(weight, bias, bn.running_mean, bn.running_var, float(bn.eps), N, scale, shift)
            mean.copy_(bn.running_mean)
            rstd.copy_(torch.rsqrt(bn.running_var.float() + float(bn.eps)))
            count = BNState.EVAL_COUNT
        Z = torch.empty(P, N, dtype=torch.float32, device=dev)
        ops.bnrelu_apply(X, P, N, scale, shift, Z, use_drop=drop is not None, salt=salt, drop=drop)
        ctx.save_for_backward(X, weight, scale, shift, mean, rstd)
        ctx.count, ctx.drop, ctx.salt = count, drop, salt
        return Z

    @staticmethod
    def backward(ctx, dZ):
        ops = _ops()
        X, weight, scale, shift, mean, rstd = ctx.saved_tensors
        P, N = X.shape
        dev = X.device
        dZ = dZ.float().contiguous()
        dz = torch.empty(P, N, dtype=torch.float32, device=dev)
        nb = ops.rowwise_blocks(P, N)
        part = torch.empty(nb, N, 2, dtype=torch.float32, device=dev)
        # ReLU mask (re-derived from X, scale, shift) + the dropout mask of the forward + the two BatchNorm-backward column sums
        ops.bnrelu_bwd_mask(dZ, X, P, N, scale, shift, ctx.drop is not None, ctx.salt, ctx.drop, dz, part)
        dgamma = torch.zeros(N, dtype=torch.float32, device=dev)
        dbeta = torch.zeros(N, dtype=torch.float32, device=dev)
        ka, kb, kc = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(3))
        ops.bn_bwd_finalize_multi([dict(partials=part, nblk=nb, col0=0, N=N, count=ctx.count, gamma=weight, mean=mean, rstd=rstd,
                                        dgamma=dgamma, dbeta=dbeta, ka=ka, kb=kb, kc=kc, accumulate=False)])
        ops.bn_bwd_apply(dz, X, P, N, ka, kb, kc)            # dz <- dx in place
        return dz, dgamma, dbeta, None, None, None


class SemchAgg(torch.autograd.Function):
    """Y (P, 2C) = [aggregate_a(h0_a, h1_a) | aggregate_b(h0_b, h1_b)] with H (P, 4C) = [h0_a | h1_a | h0_b | h1_b] and the
    channel-wise masked-softmax adjacencies of e_a (C, nnz_a), e_b (C, nnz_b)   (reference local_attention.py:40-48)."""

    @staticmethod
    def forward(ctx, H, e_a, e_b, meta):
        ops = _ops()
        (tab_a, nnz_a, dr_a, dc_a), (tab_b, nnz_b, dr_b, dc_b), F, J, C = meta
        dev = H.device
        H = H.float().contiguous()
        e_a, e_b = e_a.float().contiguous(), e_b.float().contiguous()
        A_a = torch.empty(nnz_a + 1, C, dtype=torch.float32, device=dev)
        A_b = torch.empty(nnz_b + 1, C, dtype=torch.float32, device=dev)
        ops.semch_adj_fwd(e_a, tab_a, A_a)
        ops.semch_adj_fwd(e_b, tab_b, A_b)
        Y = torch.empty(F * J, 2 * C, dtype=torch.float32, device=dev)
        part = torch.empty(ops.semch_agg_blocks(F, C), 2 * C, 2, dtype=torch.float32, device=dev)
        ops.semch_agg_fwd(H, F, J, C, A_a, tab_a, A_b, tab_b, Y, part, deg=(dr_a, dr_b))
        ctx.save_for_backward(H, A_a, A_b)
        ctx.meta = meta
        return Y

    @staticmethod
    def backward(ctx, dY):
        ops = _ops()
        H, A_a, A_b = ctx.saved_tensors
        (tab_a, nnz_a, dr_a, dc_a), (tab_b, nnz_b, dr_b, dc_b), F, J, C = ctx.meta
        dev = H.device
        dY = dY.float().contiguous()
        dH = torch.empty_like(H)
        dA = torch.empty(nnz_a + nnz_b, C, dtype=torch.float32, device=dev)
        ws = torch.empty(max(1, ops.semch_agg_bwd_ws(F, C, nnz_a, nnz_b)), dtype=torch.float32, device=dev)
        ops.semch_agg_bwd(dY, H, F, J, C, A_a, tab_a, A_b, tab_b, dH, dA, ws, cdeg=(dc_a, dc_b))
        de_a = torch.empty(C, nnz_a, dtype=torch.float32, device=dev)
        de_b = torch.empty(C, nnz_b, dtype=torch.float32, device=dev)
        ops.semch_adj_bwd(dA[:nnz_a], A_a, tab_a, de_a)
        ops.semch_adj_bwd(dA[nnz_a:], A_b, tab_b, de_b)
        return dH, de_a, de_b, None


class Attn(torch.autograd.Function):
    """Ya (P, Cg) = (softmax_j LeakyReLU_0.2(a_i + c_j) + C_k) . g per (frame, head); G (P, Cg) value projections of all heads, AC (P, 2 nh)
    = [a (heads) | c (heads)], Ck (nh, J, J)   (reference global_attention.py:60-78)."""

    @staticmethod
    def forward(ctx, G, AC, Ck, F, J, nheads):
        ops = _ops()
        Cg = G.shape[1]
        # (one buffer [G | AC] with 16-byte aligned column blocks, as the G1 output of the fused plan)
        Np = (Cg + 2 * nheads + 3) // 4 * 4              # row pitch of 16 bytes
        H = torch.empty(G.shape[0], Np, dtype=torch.float32, device=G.device)[:, :Cg + 2 * nheads]
        H[:, :Cg] = G
        H[:, Cg:] = AC
        Ck = Ck.float().contiguous()
        Ya = torch.empty(G.shape[0], Cg, dtype=torch.float32, device=G.device)
        ops.attn_fwd(H[:, :Cg], H[:, Cg:], Ck, F, J, Cg, nheads, Ya)
        ctx.save_for_backward(H, Ck)
        ctx.dims = (F, J, Cg, nheads)
        return Ya

    @staticmethod
    def backward(ctx, dYa):
        ops = _ops()
        H, Ck = ctx.saved_tensors
        F, J, Cg, nheads = ctx.dims
        dYa = dYa.float().contiguous()
        dH = torch.empty(H.shape[0], (H.shape[1] + 3) // 4 * 4, dtype=torch.float32, device=H.device)[:, :H.shape[1]]
        dCk = torch.zeros_like(Ck)                    # accumulated with atomics
        ops.attn_bwd(dYa, H[:, :Cg], H[:, Cg:], Ck, F, J, Cg, nheads, dH[:, :Cg], dH[:, Cg:], dCk)
        return dH[:, :Cg], dH[:, Cg:], dCk, None, None, None
