"""Lazy BatchNorm (include/gast_hip.h: gast_bn_lazy) at kernel level -- GPU only.

Producers accumulate their column sums into a float64 slab with atomics, consumers derive scale / shift from the slab in their own
prologue, one gast_bn_finalize_sums launch writes the tables and running statistics.  Checked here, every entry point against the
numpy contract (oracle/kernel_contract.py):
  * producers: the slab equals the column totals of the two-phase partial rows (gemm on both kernels incl. split-K, aggregation,
    input statistics, expand conv, ReLU-mask backward);
  * consumers: the lazy form equals the table form fed with gast_bn_finalize_sums's tables BIT FOR BIT (the backward pass
    re-derives ReLU masks from those tables: a one-ulp disagreement would flip decisions) and the contract within fp32 round-off;
  * gast_bn_finalize_sums == gast_bn_finalize's contract; gast_bn_bwd_apply_lazy == finalize + apply of the contract.
"""
import numpy as np
import pytest
import torch

from oracle import kernel_contract as kc
from test_kernels_gpu import (GEMM_BIG_CASES, GEMM_CASES, _gemm_case, _with_images, close, host, ops, patterns, x3_mode)  # noqa: F401

pytestmark = pytest.mark.gpu


def make_bn(gen, n, count):
    """a plausible slab {sum x, sum x^2} (+ gamma, beta) of n channels over `count` rows"""
    mean = torch.randn(n, generator=gen, dtype=torch.float64) * 0.3
    var = torch.rand(n, generator=gen, dtype=torch.float64) * 0.8 + 0.2
    sums = torch.stack([mean * count, (var + mean * mean) * count], dim=1).contiguous()
    gamma = torch.rand(n, generator=gen) + 0.5
    beta = torch.randn(n, generator=gen) * 0.3
    return sums, gamma, beta


def lazy_of(sums, gamma, beta, count, eps=1e-5):
    from gast_hip.binding import BnLazy
    return BnLazy(sums.cuda(), gamma.cuda(), beta.cuda(), float(count), eps, sums.shape[0])


def tables(ops, sums, gamma, beta, count, eps=1e-5, momentum=0.1):
    """gast_bn_finalize_sums -> device tables (scale, shift, mean, rstd) + buffers; and the contract's float64 versions"""
    n = sums.shape[0]
    d = {k: torch.zeros(n).cuda() for k in ('scale', 'shift', 'mean', 'rstd')}
    rm, rv = torch.full((n,), 0.25), torch.full((n,), 1.5)
    rmd, rvd, nbt = rm.cuda(), rv.cuda(), torch.tensor(3, dtype=torch.int64).cuda()
    ops.bn_finalize_sums([dict(sums=sums.cuda(), N=n, count=float(count), gamma=gamma.cuda(), beta=beta.cuda(), running_mean=rmd,
                               running_var=rvd, nbt=nbt, momentum=momentum, eps=eps, **d)])
    ref = {k: np.zeros(n) for k in ('scale', 'shift', 'mean', 'rstd')}
    rmh, rvh, nbh = rm.double().numpy().copy(), rv.double().numpy().copy(), np.array(3, np.int64)
    kc.bn_finalize(sums.numpy()[None], 1, 0, n, float(count), gamma.double().numpy(), beta.double().numpy(), rmh, rvh, nbh, momentum, eps,
                   ref['scale'], ref['shift'], ref['mean'], ref['rstd'])
    return d, ref, (rmd, rvd, nbt), (rmh, rvh, nbh)


def test_bn_finalize_sums_matches_contract(ops):
    gen = torch.Generator().manual_seed(1)
    for n, count in ((2, 58752), (128, 54400), (1024, 2176), (36, 7)):
        sums, gamma, beta = make_bn(gen, n, count)
        d, ref, (rmd, rvd, nbt), (rmh, rvh, nbh) = tables(ops, sums, gamma, beta, count)
        for k in ref:
            np.testing.assert_allclose(host(d[k]), ref[k], rtol=2e-6, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(host(rmd), rmh, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(host(rvd), rvh, rtol=2e-6, atol=1e-7)
        assert int(nbt.item()) == int(nbh) == 4


@pytest.mark.parametrize('use_drop', [False, True])
def test_bnrelu_apply_lazy(ops, use_drop):
    """two descriptors over adjacent column ranges (lcat_bn | gcat_bn): bit-equal to the table form, contract within round-off"""
    from gast_hip.binding import Dropout, dropout_params
    gen = torch.Generator().manual_seed(2)
    rows, n0, n1 = 1000, 64, 32
    N = n0 + n1
    X = torch.randn(rows, N + 8, generator=gen)
    bns = [make_bn(gen, n0, rows), make_bn(gen, n1, rows)]
    lz = [lazy_of(*b, rows) for b in bns]
    tabs = [tables(ops, *b, rows) for b in bns]
    sc = torch.cat([t[0]['scale'] for t in tabs])
    sh = torch.cat([t[0]['shift'] for t in tabs])
    thresh, inv_keep = dropout_params(0.2)
    drop = Dropout(torch.tensor([99], dtype=torch.int32).cuda(), thresh, inv_keep)
    Xd = X.cuda()[:, :N]
    Y_lazy, Y_tab = torch.empty(rows, N).cuda(), torch.empty(rows, N).cuda()
    ops.bnrelu_apply(Xd, rows, N, None, None, Y_lazy, use_drop=use_drop, salt=4, drop=drop, lazy=lz)
    ops.bnrelu_apply(Xd, rows, N, sc, sh, Y_tab, use_drop=use_drop, salt=4, drop=drop)
    assert torch.equal(Y_lazy, Y_tab), 'lazy coefficients differ from the gast_bn_finalize_sums tables'
    Yh = np.zeros((rows, N))
    hX = np.zeros((rows, N + 8))
    hX[:] = host(X)
    kc.bnrelu_apply(hX[:, :N], rows, N, np.concatenate([t[1]['scale'] for t in tabs]), np.concatenate([t[1]['shift'] for t in tabs]), Yh,
                    use_drop=use_drop, salt=4, drop=(99, thresh, inv_keep))
    # (a ReLU input within round-off of zero may be decided differently in float64: compare away from the kink)
    z = hX[:, :N] * np.concatenate([t[1]['scale'] for t in tabs]) + np.concatenate([t[1]['shift'] for t in tabs])
    ok = np.abs(z) > 1e-5
    assert np.abs(host(Y_lazy) - Yh)[ok].max() < 2e-5


def test_residual_fwd_lazy(ops):
    from gast_hip.binding import Dropout, dropout_params
    gen = torch.Generator().manual_seed(3)
    B, Tn, J, N, Tp = 3, 5, 17, 96, 11
    rows = B * Tn * J
    O = torch.randn(B * Tp * J, N, generator=gen)
    T2 = torch.randn(rows, N, generator=gen)
    omap = kc.RowMap(Tp, 1, 3)
    bO, b2 = make_bn(gen, N, B * Tp * J), make_bn(gen, N, rows)
    tO, t2 = tables(ops, *bO, B * Tp * J), tables(ops, *b2, rows)
    thresh, inv_keep = dropout_params(0.1)
    drop = Dropout(torch.tensor([5], dtype=torch.int32).cuda(), thresh, inv_keep)
    Xl, Xt = torch.empty(rows, N).cuda(), torch.empty(rows, N).cuda()
    ops.residual_fwd(O.cuda(), omap, None, None, T2.cuda(), None, None, True, 6, drop, B, Tn, J, N, Xl,
                     lazyO=lazy_of(*bO, B * Tp * J), lazy2=lazy_of(*b2, rows))
    ops.residual_fwd(O.cuda(), omap, tO[0]['scale'], tO[0]['shift'], T2.cuda(), t2[0]['scale'], t2[0]['shift'], True, 6, drop, B, Tn, J, N, Xt)
    assert torch.equal(Xl, Xt)
    Xh = np.zeros((rows, N))
    kc.residual_fwd(host(O), omap, tO[1]['scale'], tO[1]['shift'], host(T2), t2[1]['scale'], t2[1]['shift'], True, 6, (5, thresh, inv_keep),
                    B, Tn, J, N, Xh)
    assert np.median(np.abs(host(Xl) - Xh)) < 1e-6 and np.mean(np.abs(host(Xl) - Xh) > 1e-4) < 1e-3      # (kinks aside)


def test_input_stats_and_expand_lazy(ops):
    """init_bn statistics into a slab, expand conv reading init_bn lazily and accumulating expand_bn's slab"""
    gen = torch.Generator().manual_seed(4)
    B, T_in, J, F_in, k0, C = 5, 9, 17, 2, 3, 64
    rows_in = B * T_in * J
    x = torch.rand(B, T_in, J, F_in, generator=gen) * 2 - 1
    s0 = torch.zeros(F_in, 2, dtype=torch.float64).cuda()
    ops.input_stats(x.cuda(), rows_in, F_in, None, sums=s0)
    ph = np.zeros((kc.input_stats_blocks(rows_in), F_in, 2))
    kc.input_stats(host(x), rows_in, F_in, ph)
    np.testing.assert_allclose(s0.cpu().numpy(), ph.sum(axis=0), rtol=1e-5, atol=1e-4)
    g0, b0 = torch.rand(F_in, generator=gen) + 0.5, torch.randn(F_in, generator=gen) * 0.2
    W = torch.randn(C, F_in, k0, generator=gen) * 0.5
    T_out = T_in - k0 + 1
    P = B * T_out * J
    sE = torch.zeros(C, 2, dtype=torch.float64).cuda()
    E_l = torch.empty(P, C).cuda()
    ops.expand_fwd(x.cuda(), B, T_in, J, F_in, k0, 1, W.cuda(), None, None, C, E_l, None, lazy0=lazy_of(s0.cpu(), g0, b0, rows_in), sums=sE)
    t0 = tables(ops, s0.cpu(), g0, b0, rows_in)
    E_t = torch.empty(P, C).cuda()
    nb = ops.rowwise_blocks(P, C)
    partE = torch.zeros(nb, C, 2).cuda()
    ops.expand_fwd(x.cuda(), B, T_in, J, F_in, k0, 1, W.cuda(), t0[0]['scale'], t0[0]['shift'], C, E_t, partE)
    assert torch.equal(E_l, E_t)
    np.testing.assert_allclose(sE.cpu().numpy(), host(partE).sum(axis=0), rtol=1e-5, atol=1e-3)
    Eh, peh = np.zeros((P, C)), np.zeros((kc.rowwise_blocks(P, C), C, 2))
    kc.expand_fwd(host(x), B, T_in, J, F_in, k0, 1, host(W), t0[1]['scale'], t0[1]['shift'], C, Eh, peh)
    close(host(E_l), Eh, torch.float32, 'expand lazy')


@pytest.mark.parametrize('J', [17, 19])
def test_semch_agg_fwd_sums(ops, J):
    gen = torch.Generator().manual_seed(5)
    F, C = 37, 64
    ps, pc = patterns(J)
    P = F * J
    H = torch.randn(P, 4 * C + 8, generator=gen)
    ns, nc = int(ps[1]), int(pc[1])
    A_s = torch.rand(ns + 1, C, generator=gen)
    A_c = torch.rand(nc + 1, C, generator=gen)
    A_s[-1] = 0
    A_c[-1] = 0
    pat_s, pat_c = torch.as_tensor(ps).cuda(), torch.as_tensor(pc).cuda()
    deg = (int(ps[2 + 2 * (J + 1) + 3 * ns]), int(pc[2 + 2 * (J + 1) + 3 * nc]))
    Y1, Y2 = torch.empty(P, 2 * C).cuda(), torch.empty(P, 2 * C).cuda()
    part = torch.zeros(ops.semch_agg_blocks(F, C), 2 * C, 2).cuda()
    sums = torch.zeros(2 * C, 2, dtype=torch.float64).cuda()
    Hd = H.cuda()
    ops.semch_agg_fwd(Hd, F, J, C, A_s.cuda(), pat_s, A_c.cuda(), pat_c, Y1, part, deg=deg)
    ops.semch_agg_fwd(Hd, F, J, C, A_s.cuda(), pat_s, A_c.cuda(), pat_c, Y2, None, deg=deg, sums=sums)
    assert torch.equal(Y1, Y2)
    np.testing.assert_allclose(sums.cpu().numpy(), host(part).sum(axis=0), rtol=1e-5, atol=1e-3)


def test_bnrelu_bwd_mask_sums(ops):
    from gast_hip.binding import Dropout, dropout_params
    gen = torch.Generator().manual_seed(6)
    rows, N = 3000, 128
    dY, X = torch.randn(rows, N, generator=gen), torch.randn(rows, N, generator=gen)
    sc, sh = torch.rand(N, generator=gen) + 0.5, torch.randn(N, generator=gen) * 0.3
    thresh, inv_keep = dropout_params(0.1)
    drop = Dropout(torch.tensor([8], dtype=torch.int32).cuda(), thresh, inv_keep)
    dz1, dz2 = torch.empty(rows, N).cuda(), torch.empty(rows, N).cuda()
    part = torch.zeros(ops.rowwise_blocks(rows, N), N, 2).cuda()
    sums = torch.zeros(N, 2, dtype=torch.float64).cuda()
    ops.bnrelu_bwd_mask(dY.cuda(), X.cuda(), rows, N, sc.cuda(), sh.cuda(), True, 3, drop, dz1, part)
    ops.bnrelu_bwd_mask(dY.cuda(), X.cuda(), rows, N, sc.cuda(), sh.cuda(), True, 3, drop, dz2, None, sums=sums)
    assert torch.equal(dz1, dz2)
    np.testing.assert_allclose(sums.cpu().numpy(), host(part).sum(axis=0), rtol=1e-5, atol=1e-3)


def test_bn_bwd_apply_lazy(ops):
    """two jobs over adjacent column ranges (bn_1 | bn_2) == contract finalize + apply per job; dgamma / dbeta accumulated"""
    gen = torch.Generator().manual_seed(7)
    rows, n0, n1 = 9000, 64, 128
    N = n0 + n1
    dz = torch.randn(rows, N + 4, generator=gen) * 1e-3
    X = torch.randn(rows, N + 4, generator=gen)
    dzd = dz.cuda()
    jobs, refs, c0 = [], [], 0
    hz = np.zeros((rows, N + 4))
    hz[:] = host(dz)
    hX = host(X)
    for n in (n0, n1):
        d, x = hz[:, c0:c0 + n], hX[:, c0:c0 + n]
        sums = torch.from_numpy(np.stack([d.sum(0), (d * x).sum(0)], axis=1).copy())
        gamma = torch.rand(n, generator=gen) + 0.5
        mean, rstd = torch.randn(n, generator=gen) * 0.2, torch.rand(n, generator=gen) + 0.7
        dg0, db0 = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
        dgd, dbd = dg0.cuda(), db0.cuda()
        jobs.append(dict(sums=sums.cuda(), col0=c0, n=n, count=float(rows), gamma=gamma.cuda(), mean=mean.cuda(), rstd=rstd.cuda(),
                         dgamma=dgd, dbeta=dbd))
        ka, kb, kcc = np.zeros(n), np.zeros(n), np.zeros(n)
        dgh, dbh = host(dg0).copy(), host(db0).copy()
        kc.bn_bwd_finalize(sums.numpy()[None], 1, 0, n, float(rows), host(gamma), host(mean), host(rstd), dgh, dbh, ka, kb, kcc, accumulate=True)
        refs.append((c0, n, ka, kb, kcc, dgh, dbh, dgd, dbd))
        c0 += n
    ops.bn_bwd_apply_lazy(dzd[:, :N], X.cuda()[:, :N], rows, jobs)
    got = host(dzd)
    for c0, n, ka, kb, kcc, dgh, dbh, dgd, dbd in refs:
        ref = hz[:, c0:c0 + n].copy()
        kc.bn_bwd_apply(ref, hX[:, c0:c0 + n], rows, n, ka, kb, kcc)
        close(got[:, c0:c0 + n], ref, torch.float32, 'bn_bwd_apply_lazy dx')
        np.testing.assert_allclose(host(dgd), dgh, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(host(dbd), dbh, rtol=1e-5, atol=1e-6)
    assert np.array_equal(got[:, N:], hz[:, N:]), 'wrote outside the N columns'


PRO_BIG = [c for c in GEMM_BIG_CASES if any(sd[4] == 1 for sd in c[3])]


@pytest.mark.parametrize('pair', ['bf16', 'f16'])
@pytest.mark.parametrize('case', PRO_BIG, ids=[c[0] for c in PRO_BIG])
def test_gemm_big_lazy_segments(ops, case, pair):
    """the large-M kernel with its BatchNorm prologue segments read lazily (+ the statistics epilogue into a slab): bit-equal to the
    same GEMM fed with gast_bn_finalize_sums's tables, slab == column totals of the partial rows"""
    gen = torch.Generator().manual_seed(11)
    jd, jh, bufs = _gemm_case(case, torch.float32)
    f16 = pair == 'f16'
    rows = 4321.0
    lazy_segs = []
    for sg in jd['segs']:
        if sg['pro'] == 1:
            bn = make_bn(gen, sg['K'], rows)
            t = tables(ops, *bn, rows)
            sg['scale'], sg['shift'] = t[0]['scale'], t[0]['shift']
            lazy_segs.append((sg, lazy_of(*bn, rows)))
    N = case[2]
    with x3_mode(ops, 'x3'):
        _with_images(ops, jd, f16)
        assert ops.gemm_path(**jd) == 1
        ops.gemm(**jd)                                   # table form
        C_tab, part = bufs[0].clone(), bufs[2].clone() if bufs[2] is not None else None
        bufs[0].fill_(7.0)
        for sg, lz in lazy_segs:
            sg['lazy'] = lz
        sums = torch.zeros(N, 2, dtype=torch.float64).cuda() if case[4] else None
        jd2 = dict(jd, partials=None, stat_sums=sums) if case[4] else jd
        assert ops.gemm_path(**jd2) == 1
        ops.gemm(**jd2)
    torch.cuda.synchronize()
    assert torch.equal(bufs[0], C_tab), 'lazy prologue coefficients differ from the tables'
    if case[4]:
        np.testing.assert_allclose(sums.cpu().numpy(), host(part).sum(axis=0), rtol=2e-5, atol=2e-3)


STAT_SMALL = [c for c in GEMM_CASES if c[4]]


@pytest.mark.parametrize('mode', ['f32', 'x3'])
@pytest.mark.parametrize('case', STAT_SMALL, ids=[c[0] for c in STAT_SMALL])
def test_gemm_small_stat_sums(ops, case, mode):
    """the 128x128-tile kernel (and its split-K finish) accumulating its statistics into a slab; a lazy K segment is REJECTED there"""
    jd, jh, bufs = _gemm_case(case, torch.float32)
    N = case[2]
    with x3_mode(ops, mode):
        ops.gemm(**jd)
        C1, part = bufs[0].clone(), bufs[2].clone()
        bufs[0].fill_(7.0)
        sums = torch.zeros(N, 2, dtype=torch.float64).cuda()
        ops.gemm(**dict(jd, partials=None, stat_sums=sums))
        torch.cuda.synchronize()
        assert torch.equal(bufs[0], C1)
        np.testing.assert_allclose(sums.cpu().numpy(), host(part).sum(axis=0), rtol=2e-5, atol=2e-3)
        pro = [sg for sg in jd['segs'] if sg['pro'] == 1]
        if pro and ops.gemm_path(**jd) == 0:
            gen = torch.Generator().manual_seed(1)
            pro[0]['lazy'] = lazy_of(*make_bn(gen, pro[0]['K'], 100.0), 100.0)
            with pytest.raises(RuntimeError, match='GAST_EINVAL'):
                ops.gemm(**jd)
