"""numpy mirror of `gast_hip.binding.HipOps` on CPU torch tensors -- TEST INFRASTRUCTURE ONLY.

It lets the product's host-side plan (gast_hip/engine.py) run on CPU so that its *composition* of ops can be pinned
against the reference-generated golden fixtures without a GPU (tests/test_plan_cpu.py).  Every op forwards to
oracle/kernel_contract.py.  It lives under tests/ and is handed to a model explicitly (`model._runner.set_ops(OracleOps)`, see
use_oracle_ops below); nothing in the product imports it or looks it up.
"""
import numpy as np
import torch

from oracle import kernel_contract as kc


def _np(t):
    if t is None:
        return None
    assert not t.is_cuda
    return t.detach().numpy()


def _segs(segs, a_key):
    out = []
    for s in segs:
        d = dict(s)
        d[a_key] = _np(s[a_key])
        if 'W' in d:
            d['W'] = _np(d['W'])
        d['scale'] = _np(s.get('scale'))
        d['shift'] = _np(s.get('shift'))
        d['map'] = kc.RowMap(*s['map'])
        out.append(d)
    return out


def _drop(d):
    if d is None:
        return None
    return (int(d.seed.item()) & 0xffffffff, int(d.thresh), float(d.inv_keep))


class OracleOps:
    name = 'oracle-mirror'

    def __init__(self):
        self.launches = 0

    def gemm_row_blocks(self, M):
        return kc.gemm_row_blocks(M)

    def prep(self, zero, seed=None, pad=None):
        self.launches += 1
        kc.prep([_np(t) for t in zero], seed=None if seed is None else (_np(seed[0]), _np(seed[1])),
                pad=None if pad is None else (_np(pad[0]), _np(pad[1])) + tuple(pad[2:]))

    def gemm(self, dom, N, segs, C_, cmap, bias=None, addend=None, addmap=None, epi=0, partials=None, X=None, xscale=None,
             xshift=None, xdrop=False, xsalt=0, drop=None, bias_neg=False, C2=None):
        self.launches += 1
        if C2 is not None:        # second output of the BNRELU_BWD epilogue: the value before the mask = the same GEMM with the PLAIN epilogue
            assert epi == 2
            kc.gemm(dom, N, _segs(segs, 'A'), _np(C2), kc.RowMap(*cmap), _np(bias), _np(addend),
                    kc.RowMap(*addmap) if addmap is not None else None, 0, None, None, None, None, False, 0, None, bias_neg=bias_neg)
        kc.gemm(dom, N, _segs(segs, 'A'), _np(C_), kc.RowMap(*cmap), _np(bias), _np(addend),
                kc.RowMap(*addmap) if addmap is not None else None, epi, _np(partials), _np(X), _np(xscale), _np(xshift),
                xdrop, xsalt, _drop(drop), bias_neg=bias_neg)

    def gemm_multi(self, jobs):
        for j in jobs:
            self.gemm(**j)

    def wgrad(self, dom, P, R, pmap, segs, dW, drop=None, zero_first=True):
        self.launches += 1
        kc.wgrad(dom, _np(P), R, kc.RowMap(*pmap), _segs(segs, 'Q'), _np(dW), _drop(drop), zero_first)

    def wgrad_multi(self, jobs):
        for j in jobs:
            self.wgrad(**j)

    def semch_adj_fwd(self, e, pat, A_t):
        kc.semch_adj_fwd(_np(e), _np(pat), _np(A_t))

    def semch_adj_bwd(self, dA_t, A_t, pat, de):
        kc.semch_adj_bwd(_np(dA_t), _np(A_t), _np(pat), _np(de))

    def semch_agg_blocks(self, F, C_):
        return kc.semch_agg_blocks(F, C_)

    def semch_adj_fwd_multi(self, jobs):
        for e, pat, A_t in jobs:
            self.semch_adj_fwd(e, pat, A_t)

    def semch_adj_bwd_multi(self, jobs, accumulate=False):
        for dA_t, A_t, pat, de in jobs:
            kc.semch_adj_bwd(_np(dA_t), _np(A_t), _np(pat), _np(de), accumulate=accumulate)

    def semch_agg_fwd(self, H, F, J, C_, A_sym, pat_sym, A_con, pat_con, Y, partials, deg=(0, 0), center=(None, None)):
        kc.semch_agg_fwd(_np(H), F, J, C_, _np(A_sym), _np(pat_sym), _np(A_con), _np(pat_con), _np(Y), _np(partials),
                         center_sym=_np(center[0]), center_con=_np(center[1]))

    def semch_agg_bwd_ws(self, F, C_, nnz_sym, nnz_con):
        return 1

    def semch_agg_bwd(self, dY, H, F, J, C_, A_sym, pat_sym, A_con, pat_con, dH, dA, ws, cdeg=(0, 0)):
        dAn = _np(dA)
        dAn[...] = 0
        ns = A_sym.shape[0] - 1
        kc.semch_agg_bwd(_np(dY), _np(H), F, J, C_, _np(A_sym), _np(pat_sym), _np(A_con), _np(pat_con), _np(dH), dAn[:ns], dAn[ns:])

    def attn_fwd(self, G, AC, C_k, F, J, C_, nheads, Y):
        kc.attn_fwd(_np(G), _np(AC), _np(C_k), F, J, C_, nheads, _np(Y))

    def attn_bwd(self, dY, G, AC, C_k, F, J, C_, nheads, dG, dAC, dC_k, dbias=None, generic=False):
        kc.attn_bwd(_np(dY), _np(G), _np(AC), _np(C_k), F, J, C_, nheads, _np(dG), _np(dAC), _np(dC_k), dbias=_np(dbias))

    def bn_finalize(self, partials, nblk, col0, N, count, gamma, beta, running_mean, running_var, nbt, momentum, eps, scale,
                    shift, mean, rstd, centered=False):
        kc.bn_finalize(_np(partials), nblk, col0, N, count, _np(gamma), _np(beta), _np(running_mean), _np(running_var), _np(nbt),
                       momentum, eps, _np(scale), _np(shift), _np(mean), _np(rstd), centered=centered)

    def bn_finalize_multi(self, jobs):
        for j in jobs:
            self.bn_finalize(**j)

    def bn_bwd_finalize_multi(self, jobs):
        for j in jobs:
            j = dict(j)
            j['kc_'] = j.pop('kc')
            self.bn_bwd_finalize(**j)

    def bn_bwd_fused_multi(self, jobs):
        import numpy as np
        for j in jobs:
            N = j['N']
            ka, kb, kc_ = np.zeros(N), np.zeros(N), np.zeros(N)
            kc.bn_bwd_finalize(_np(j['partials']), j['nblk'], j['col0'], N, j['count'], _np(j['gamma']), _np(j['mean']), _np(j['rstd']),
                               _np(j['dgamma']), _np(j['dbeta']), ka, kb, kc_, accumulate=j.get('accumulate', False))
            kc.bn_bwd_apply(_np(j['dz']), _np(j['X']), j['rows'], N, ka, kb, kc_)

    def bn_eval_multi(self, jobs, eps):
        for gamma, beta, rm, rv, scale, shift, centered in jobs:
            self.bn_eval(gamma, beta, rm, rv, eps, gamma.numel(), scale, shift, centered=centered)

    def bn_eval(self, gamma, beta, rm, rv, eps, N, scale, shift, centered=False):
        kc.bn_eval(_np(gamma), _np(beta), _np(rm), _np(rv), eps, N, _np(scale), _np(shift), centered=centered)

    def bn_bwd_finalize(self, partials, nblk, col0, N, count, gamma, mean, rstd, dgamma, dbeta, ka, kb, kc_, accumulate=False):
        kc.bn_bwd_finalize(_np(partials), nblk, col0, N, count, _np(gamma), _np(mean), _np(rstd), _np(dgamma), _np(dbeta), _np(ka),
                           _np(kb), _np(kc_), accumulate=accumulate)

    def bn_bwd_apply(self, dz, X, rows, N, ka, kb, kc_):
        kc.bn_bwd_apply(_np(dz), _np(X), rows, N, _np(ka), _np(kb), _np(kc_))

    def bnrelu_apply(self, X, rows, N, scale, shift, Y, use_drop=False, salt=0, drop=None):
        kc.bnrelu_apply(_np(X), rows, N, _np(scale), _np(shift), _np(Y), use_drop=use_drop, salt=salt, drop=_drop(drop))

    def rowwise_blocks(self, rows, N):
        return kc.rowwise_blocks(rows, N)

    def bnrelu_bwd_mask(self, dY, X, rows, N, scale, shift, use_drop, salt, drop, dz, partials):
        kc.bnrelu_bwd_mask(_np(dY), _np(X), rows, N, _np(scale), _np(shift), use_drop, salt, _drop(drop), _np(dz), _np(partials))

    def residual_fwd(self, O, omap, scO, shO, T2, sc2, sh2, use_drop, salt, drop, B, Tn, J, N, Xn):
        kc.residual_fwd(_np(O), kc.RowMap(*omap), _np(scO), _np(shO), _np(T2), _np(sc2), _np(sh2), use_drop, salt, _drop(drop),
                        B, Tn, J, N, _np(Xn))

    def colsum(self, X, rows, N, out, zero_first=True):
        kc.colsum(_np(X), rows, N, _np(out), zero_first)

    def input_stats_blocks(self, rows):
        return kc.input_stats_blocks(rows)

    def input_stats(self, x, rows, F_in, partials):
        kc.input_stats(_np(x), rows, F_in, _np(partials))

    def expand_fwd(self, x, B, T_in, J, F_in, k0, t_stride, W, sc0, sh0, C_, E, partials, center=None):
        kc.expand_fwd(_np(x), B, T_in, J, F_in, k0, t_stride, _np(W), _np(sc0), _np(sh0), C_, _np(E), _np(partials), center=_np(center))

    def expand_bwd(self, dE, x, B, T_in, J, F_in, k0, t_stride, mean0, rstd0, C_, W, gamma0, beta0, dW, dgamma0, dbeta0, accumulate=False):
        kc.expand_bwd(_np(dE), _np(x), B, T_in, J, F_in, k0, t_stride, _np(mean0), _np(rstd0), C_, _np(W), _np(gamma0), _np(beta0),
                      _np(dW), _np(dgamma0), _np(dbeta0), accumulate=accumulate)


def _resolve(ref, bases, R, S):
    from gast_hip.packer import BASE_ABS
    flat = ref.tensor.detach().view(-1) if ref.base == BASE_ABS else bases[ref.base]
    return torch.as_strided(flat, (R, S), (ref.rs, ref.cs), ref.off)


def _run_pack(self, packer, st):
    from gast_hip.packer import BASE_W, BASE_F
    bases = {BASE_W: st['Wb'], BASE_F: st['Fb']}
    for job in packer.copy_jobs:
        src, dst, R, S = job[:4]
        _resolve(dst, bases, R, S).copy_(_resolve(src, bases, R, S))
    for j in packer.fold_jobs:
        Ci, Cc = j['Ci'], j['C']
        W = j['W'].detach().view(Ci, Cc).double()
        w = j['w'].detach().view(-1)[j['woff']:j['woff'] + Ci].double()
        v = (W * w[:, None]).sum(0)
        torch.as_strided(bases[BASE_W], (Cc,), (j['row'].cs,), j['row'].off).copy_(v)
        torch.as_strided(bases[BASE_W], (Cc,), (j['col'].cs,), j['col'].off).copy_(v)
        bases[BASE_F][j['bias'].off] = float((w * j['b'].detach().double()).sum())


def _run_unpack(self, packer, st, Sb, G, accumulate, bucket=None):
    from gast_hip.packer import BASE_S, BASE_G
    bases = {BASE_S: Sb, BASE_G: G}
    for src, dst, R, S in (packer.unpack_jobs if bucket is None else packer.unpack_by_bucket[bucket]):
        d = _resolve(dst, bases, R, S)
        v = _resolve(src, bases, R, S)
        d.add_(v) if accumulate else d.copy_(v)
    for j in (packer.unfold_jobs if bucket is None else packer.unfold_by_bucket[bucket]):
        Ci, Cc = j['Ci'], j['C']
        dv = torch.as_strided(Sb, (Cc,), (1,), j['dv'].off).double()
        da = float(Sb[j['da'].off])
        W = j['W'].detach().view(Ci, Cc).double()
        w = j['w'].detach().view(-1)[j['woff']:j['woff'] + Ci].double()
        b = j['b'].detach().double()
        gW = packer.goff[packer.index[id(j['W'])]]
        gw = packer.goff[packer.index[id(j['w'])]] + j['woff']
        gb = packer.goff[packer.index[id(j['b'])]]
        outs = ((G[gW:gW + Ci * Cc].view(Ci, Cc), w[:, None] * dv[None, :]), (G[gw:gw + Ci], W @ dv + b * da), (G[gb:gb + Ci], w * da))
        for dst, val in outs:
            dst.add_(val.to(dst.dtype)) if accumulate else dst.copy_(val.to(dst.dtype))


OracleOps.run_pack = _run_pack
OracleOps.run_unpack = _run_unpack


def use_oracle_ops(model):
    """Route a model instance through the numpy mirror (CPU tensors allowed)."""
    model._runner.set_ops(OracleOps)
    return model
