"""Every `gast_*` C-ABI kernel against its numpy contract (oracle/kernel_contract.py) on seeded inputs -- GPU only.

fp32: tight tolerances (MFMA fp32 is an exact fmaf chain; only summation order differs from numpy/float64).
bf16: inputs are rounded to bf16 first, the contract computes in float64 on those rounded inputs, outputs are compared
with a tolerance of a few bf16 ulps of the result magnitude.
"""
import numpy as np
import pytest
import torch

from oracle import kernel_contract as kc
from tests_helpers import PARENTS

pytestmark = pytest.mark.gpu

# The 16-bit STORAGE flavour under test (csrc/common.h: the library is built twice): bfloat16, or -- GAST_TEST_H16=f16, set by
# tests/test_f16_gpu.py for a child run of the "bf16" parametrisations of this file -- IEEE binary16 through libgast_hip_f16.so.
import os as _os
H16 = torch.float16 if _os.environ.get('GAST_TEST_H16') == 'f16' else torch.bfloat16
DTYPES = [torch.float32, H16]
# MFMA kernels (gast_gemm*, gast_wgrad*): fp32, bf16 and fp32 storage with split-bf16 products (GAST_F32X3); 'x3' runs the fp32
# cases with HipOps.x3 set and a tolerance of 1e-4 of the result magnitude (three bf16 products carry ~2^-17 relative each)
# 'x3h' (GAST_F32X3H): the same with fp16 hi/lo pairs -- the weight operands carry the tag (X3Weight.f16), the tolerance is fp32's
MM_MODES = ['f32', 'bf16', 'x3', 'x3h']
MM_DT = {'f32': torch.float32, 'bf16': H16, 'x3': torch.float32, 'x3h': torch.float32}


class x3_mode:
    """with x3_mode(ops, mode): fp32 operands go through the split-bf16 MFMA path for the duration of the block"""

    def __init__(self, ops, mode):
        self.ops, self.on = ops, mode in ('x3', 'x3h')

    def __enter__(self):
        self.ops.x3 = self.on

    def __exit__(self, *exc):
        self.ops.x3 = False


@pytest.fixture(scope='module')
def ops():
    import os
    os.environ.setdefault('GAST_GEMM_BIG_ALL', '1')     # kernel tests: every eligible shape on the large-M kernel (read once by the library)
    os.environ.setdefault('GAST_GEMM_BJ_ALL', '1')      # ... and every eligible small-M shape on the M = B*J kernel (the product uses it where it wins)
    from conftest import poison_allocations
    from gast_hip.binding import HipOps, set_h16
    poison_allocations()           # (GAST_TEST_POISON=1 only)
    set_h16(H16)
    return HipOps()


def dev(t, dt=None):
    t = torch.as_tensor(t)
    if dt is not None:
        t = t.to(dt)
    return t.cuda()


def host(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def rand(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def tol(dt, ref, fp32=2e-5, bf16=2e-2):
    """bf16: the bound for bfloat16 storage; binary16 carries three more significand bits and is held to a quarter of it"""
    mag = max(1e-6, float(np.abs(ref).max()))
    return (fp32 if dt == torch.float32 else bf16 / 4 if dt == torch.float16 else bf16) * mag


def close(a, ref, dt, what, fp32=2e-5, bf16=2e-2):
    err = float(np.abs(a - ref).max())
    assert err <= tol(dt, ref, fp32, bf16), '%s: max err %.3e (ref max %.3e)' % (what, err, np.abs(ref).max())


def patterns(J):
    from oracle.gast_oracle import adj_from_parents, local_graph_adjacencies
    sym, con = local_graph_adjacencies(adj_from_parents(PARENTS[J]))
    return kc.build_pattern(sym), kc.build_pattern(con)


def seed_tensor(v):
    return torch.tensor([v], dtype=torch.int32).cuda()


# ------------------------------------------------------------------------------------------------ pass prologue
def test_prep_matches_contract(ops):
    """gast_prep: several zero fills (one longer than a block's share, one not 16-byte granular -> torch fill), the seed bump and
    the 3 -> 8 column padding of d loss / d pred in one call; everything outside the regions untouched."""
    gen = torch.Generator().manual_seed(3)
    big = torch.randn(300001, generator=gen)             # 1.2 MB: many blocks, ragged last block
    bufs = [big[:300000], torch.randn(4096, generator=gen), torch.randn(7, generator=gen), torch.randn(64, generator=gen),
            torch.randn(128, generator=gen), torch.randn(256, generator=gen), torch.randn(512, generator=gen), torch.randn(1024, generator=gen)]
    dev_big = big.cuda()
    dbufs = [dev_big[:300000]] + [b.cuda() for b in bufs[1:]]
    src = torch.randn(2176, 3, generator=gen)
    ctr = torch.tensor([2 ** 31 - 1], dtype=torch.int32)     # wraps like the uint32 the kernels read
    ctr_d, copy_d = ctr.cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    dst_d = torch.full((2176, 8), 7.0).cuda()
    ops.prep(dbufs, seed=(ctr_d, copy_d), pad=(src.cuda(), dst_d, 2176, 3, 8))
    ref_bufs = [b.numpy().copy() for b in bufs]
    rc, rcopy, rdst = ctr.numpy().copy(), np.zeros(1, np.int32), np.full((2176, 8), 7.0, np.float32)
    kc.prep(ref_bufs, seed=(rc, rcopy), pad=(src.numpy(), rdst, 2176, 3, 8))
    for d, r in zip(dbufs, ref_bufs):
        assert np.array_equal(d.cpu().numpy(), r)
    assert float(dev_big[300000]) == float(big[300000]), 'the element past the zeroed region changed'
    assert int(ctr_d.item()) == int(rc[0]) and int(copy_d.item()) == int(rcopy[0])
    assert np.array_equal(dst_d.cpu().numpy(), rdst)


def test_prep_pads_scaled_into_16_bit_storage(ops):
    """gast_prep's pad job with a loss scale and a 16-bit destination (GAST_HIP_DTYPE=f16 / bf16: d loss / d pred arrives in fp32, the
    shrink layer's gradient kernels read the storage type): scale * src rounded to the storage type, zero padding, nothing else."""
    gen = torch.Generator().manual_seed(5)
    src = torch.randn(301, 3, generator=gen) * 1e-4
    dst = torch.full((301, 8), 7.0).to(H16).cuda()
    ops.prep([], pad=(src.cuda(), dst, 301, 3, 8, 4096.0))
    ref = np.zeros((301, 8), np.float32)
    kc.prep([], pad=(src.numpy(), ref, 301, 3, 8, 4096.0))
    assert np.array_equal(dst.float().cpu().numpy(), torch.from_numpy(ref).to(H16).float().numpy())
    dst32 = torch.full((301, 8), 7.0).cuda()
    ops.prep([], pad=(src.cuda(), dst32, 301, 3, 8, 0.5))
    kc.prep([], pad=(src.numpy(), ref, 301, 3, 8, 0.5))
    assert np.array_equal(dst32.cpu().numpy(), ref)


# ------------------------------------------------------------------------------------------------ dropout stream
def test_dropout_stream_matches_contract(ops):
    """bnrelu_bwd_mask with scale=1, shift=1 (always positive) exposes the keep mask exactly."""
    from gast_hip.binding import Dropout, dropout_params
    rows, N = 300, 64
    thresh, inv_keep = dropout_params(0.25)
    dY = torch.ones(rows, N).cuda()
    X = torch.zeros(rows, N).cuda()
    sc = torch.ones(N).cuda()
    sh = torch.ones(N).cuda()
    dz = torch.empty(rows, N).cuda()
    part = torch.empty(ops.rowwise_blocks(rows, N), N, 2).cuda()
    ops.bnrelu_bwd_mask(dY, X, rows, N, sc, sh, True, 7, Dropout(seed_tensor(12345), thresh, inv_keep), dz, part)
    e = np.arange(rows)[:, None] * N + np.arange(N)[None, :]
    ref = kc.drop_mul(kc.drop_key(12345, 7), thresh, inv_keep, e)
    got = host(dz)
    assert np.array_equal(got != 0, ref != 0)
    frac = (got == 0).mean()
    assert abs(frac - 0.25) < 0.02
    np.testing.assert_allclose(got[got != 0], inv_keep, rtol=1e-6)


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_CASES = [
    # name, (B,Tn,J), N, list of (K, T_total, t_stride, t_off, pro), epi, addend?, bias?
    ('plain_small', (2, 3, 17), 40, [(16, 3, 1, 0, 0)], 0, False, True),
    ('stats_two_tiles', (3, 5, 17), 200, [(32, 5, 1, 0, 1)], 1, False, False),
    ('concat3_drop', (2, 7, 17), 96, [(32, 7, 1, 0, 0), (32, 7, 1, 0, 2), (32, 7, 1, 0, 2)], 1, False, False),
    ('dilated_taps', (2, 5, 17), 64, [(64, 11, 1, 0, 1), (64, 11, 1, 3, 1), (64, 11, 1, 6, 1)], 1, False, False),
    ('strided_taps', (3, 3, 19), 48, [(24, 9, 3, 0, 1), (24, 9, 3, 1, 1), (24, 9, 3, 2, 1)], 1, False, False),
    ('dgrad_gather_addend_bwd', (2, 11, 17), 64, [(64, 5, 1, 0, 0), (64, 5, 1, -3, 0), (64, 5, 1, -6, 0)], 2, True, False),
    ('ktail', (1, 9, 15), 33 * 4, [(5 * 32 + 8, 9, 1, 0, 0), (72, 9, 1, 0, 0)], 0, False, False),
    ('big', (8, 9, 17), 256, [(256, 9, 1, 0, 1)], 1, False, False),
    ('splitk_stats', (4, 1, 17), 200, [(512, 1, 1, 0, 1), (512, 1, 1, 0, 2)], 1, False, True),
    ('splitk_bwd_taps', (3, 1, 17), 64, [(256, 3, 1, 0, 0), (256, 3, 1, 1, 0), (256, 3, 1, 2, 0)], 2, True, False),
    ('splitk_plain', (2, 2, 19), 36, [(1288, 2, 1, 0, 0)], 0, False, False),
    ('centred_stats', (3, 5, 17), 136, [(64, 5, 1, 0, 1)], 1, False, 'neg'),
    ('centred_splitk', (4, 1, 17), 200, [(512, 1, 1, 0, 1), (512, 1, 1, 0, 2)], 1, False, 'neg'),
]


def _gemm_case(case, dt):
    """device keyword arguments of ops.gemm / host arguments of kc.gemm and the buffers to compare for one GEMM_CASES entry"""
    from gast_hip.binding import Dropout, dropout_params
    name, dom, N, segdefs, epi, use_add, use_bias = case
    bias_neg = use_bias == 'neg'      # centred storage: C = acc - bias
    use_bias = bool(use_bias)
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    B, Tn, J = dom
    M = B * Tn * J
    thresh, inv_keep = dropout_params(0.1)
    seed = 777
    segs_d, segs_h = [], []
    for si, (K, Tt, ts, toff, pro) in enumerate(segdefs):
        rowsA = B * Tt * J
        lda = K + 8   # exercise lda != K
        A = rand(gen, rowsA, lda).to(dt)
        W = (rand(gen, N, K) / np.sqrt(K)).to(dt)
        sc = torch.rand(K, generator=gen) + 0.5
        sh = rand(gen, K, scale=0.3)
        d = dict(A=A.cuda()[:, :K], K=K, map=kc.RowMap(Tt, ts, toff), W=W.cuda(), pro=pro, scale=sc.cuda(), shift=sh.cuda(), salt=si + 1)
        h = dict(A=host(A)[:, :K], K=K, map=kc.RowMap(Tt, ts, toff), W=host(W), pro=pro, scale=host(sc), shift=host(sh), salt=si + 1)
        # host view must have the same row stride as the device view for the dropout element index
        hA = np.zeros((rowsA, lda))
        hA[:] = host(A)
        h['A'] = hA[:, :K]
        segs_d.append(d)
        segs_h.append(h)
    cT = Tn + 2
    cmap = kc.RowMap(cT, 1, 1)
    ldc = N + 4
    Cd = torch.full((B * cT * J, ldc), 7.0).to(dt).cuda()
    Ch = np.full((B * cT * J, ldc), 7.0)
    bias = rand(gen, N) if use_bias else None
    add = rand(gen, B * (Tn + 1) * J, N).to(dt) if use_add else None
    addmap = kc.RowMap(Tn + 1, 1, 0) if use_add else None
    nb = (M + 127) // 128
    pd = torch.zeros(nb, N, 2).cuda() if epi else None
    ph = np.zeros((nb, N, 2)) if epi else None
    X = rand(gen, B * cT * J, N).to(dt) if epi == 2 else None
    xs = (torch.rand(N, generator=gen) + 0.5) if epi == 2 else None
    xh = rand(gen, N, scale=0.3) if epi == 2 else None
    rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
    jd = dict(dom=dom, N=N, segs=segs_d, C_=Cd[:, :N], cmap=cmap, bias=bias.cuda() if use_bias else None,
              addend=add.cuda() if use_add else None, addmap=addmap, epi=epi, partials=pd, X=X.cuda() if X is not None else None,
              xscale=xs.cuda() if xs is not None else None, xshift=xh.cuda() if xh is not None else None,
              xdrop=epi == 2, xsalt=9, drop=Dropout(seed_tensor(seed), thresh, inv_keep), bias_neg=bias_neg)
    jh = dict(dom=dom, N=N, segs=segs_h, C=Ch[:, :N], cmap=cmap, bias=host(bias) if use_bias else None,
              addend=host(add) if use_add else None, addmap=addmap, epi=epi, partials=ph, X=host(X) if X is not None else None,
              xscale=host(xs) if xs is not None else None, xshift=host(xh) if xh is not None else None, xdrop=epi == 2, xsalt=9,
              drop=(seed, thresh, inv_keep), round_fn=rnd, bias_neg=bias_neg)
    return jd, jh, (Cd, Ch, pd, ph)


def _f16_pairs(jd):
    """tag the weight operands of a gemm() job: its products run on fp16 pairs (what the packer does for the forward operands)"""
    from gast_hip.packer import X3Weight
    for s in jd['segs']:
        s['W'] = X3Weight(s['W'], None, True)
    return jd


def _gemm_check(case, dt, bufs, mode='f32'):
    name, N, epi = case[0], case[2], case[4]
    Cd, Ch, pd, ph = bufs
    got = host(Cd)
    close(got[:, :N], Ch[:, :N], dt, name + ' C', fp32=1e-4 if mode == 'x3' else 2e-5)
    assert np.all(got[:, N:] == 7.0), 'wrote outside the N columns'
    if epi:
        close(host(pd).sum(axis=0), ph.sum(axis=0), dt, name + ' partial totals', fp32=2e-4 if mode == 'x3' else 1e-4, bf16=3e-2)


@pytest.mark.parametrize('mode', MM_MODES)
@pytest.mark.parametrize('case', GEMM_CASES, ids=[c[0] for c in GEMM_CASES])
def test_gemm(ops, case, mode):
    dt = MM_DT[mode]
    jd, jh, bufs = _gemm_case(case, dt)
    if mode == 'x3h':
        _f16_pairs(jd)
    with x3_mode(ops, mode):
        ops.gemm(**jd)
    kc.gemm(**jh)
    torch.cuda.synchronize()
    _gemm_check(case, dt, bufs, mode)


@pytest.mark.parametrize('mode', MM_MODES)
def test_gemm_multi(ops, mode):
    """Independent GEMMs with different domains, segment counts, prologues and epilogues as multi-job launches (3 per grid).  The
    split-K-eligible ones (the M = B*J stage's shapes) ride in the same grid with their own slice of the workspace and share ONE
    finish launch (round 3; three of them are consecutive in GEMM_CASES, i.e. one call of three split-K jobs)."""
    dt = MM_DT[mode]
    cases = [c for c in GEMM_CASES]
    built = [_gemm_case(c, dt) for c in cases]
    if mode == 'x3h':
        for jd, _, _ in built:
            _f16_pairs(jd)
    with x3_mode(ops, mode):
        ops.gemm_multi([jd for jd, _, _ in built])
    torch.cuda.synchronize()
    for c, (jd, jh, bufs) in zip(cases, built):
        kc.gemm(**jh)
        _gemm_check(c, dt, bufs, mode)


# large-M cases: with pre-split weight images (HipOps.x3_weight) GAST_F32X3 GEMMs of >= 8192 rows take the large-M kernel
# (csrc/gemm_big.hip): taps + prologue + statistics + centred bias + N tail; channel concat; gather input gradient with zero rows,
# addend, ReLU/dropout/BN-sum epilogue; K tails (168 = 5*32 + 8 and 72); a one-K-tile GEMM
GEMM_BIG_CASES = [
    ('big_taps_pro_stats', (24, 21, 17), 320, [(64, 27, 1, 0, 1), (64, 27, 1, 3, 1), (64, 27, 1, 6, 1)], 1, False, 'neg'),
    ('big_concat_plain_stats', (30, 17, 17), 256, [(128, 17, 1, 0, 0), (256, 17, 1, 0, 0)], 1, False, False),
    ('big_dgrad_gather_bwd', (20, 25, 17), 128, [(64, 19, 1, 0, 0), (64, 19, 1, -3, 0), (64, 19, 1, -6, 0)], 2, True, False),
    ('big_ktail', (32, 16, 17), 136, [(5 * 32 + 8, 16, 1, 0, 0), (72, 16, 1, 0, 1)], 0, False, True),
    ('big_one_tile', (31, 16, 17), 648, [(32, 16, 1, 0, 1)], 1, False, True),
    ('big_strided_taps', (90, 5, 19), 96, [(32, 15, 3, 0, 1), (32, 15, 3, 1, 1), (32, 15, 3, 2, 1)], 1, False, False),
    ('big_bwd_noadd', (22, 23, 17), 512, [(256, 23, 1, 0, 0)], 2, False, False),
    ('big_plain_add', (22, 23, 17), 256, [(256, 23, 1, 0, 0)], 0, True, False),
]


def _with_images(ops, jd, f16=False):
    for s in jd['segs']:
        s['W'] = ops.x3_weight(s['W'], f16)
    return jd


PAIRS = ['bf16', 'f16']      # GAST_F32X3 / GAST_F32X3H


@pytest.mark.parametrize('pair', PAIRS)
@pytest.mark.parametrize('nodrop', [False, True], ids=['xdrop', 'noxdrop'])
@pytest.mark.parametrize('case', GEMM_BIG_CASES, ids=[c[0] for c in GEMM_BIG_CASES])
def test_gemm_big_x3(ops, case, nodrop, pair):
    if nodrop and case[4] != 2:
        pytest.skip('only the BNRELU_BWD epilogue has a dropout variant')
    jd, jh, bufs = _gemm_case(case, torch.float32)
    if nodrop:
        jd['xdrop'] = jh['xdrop'] = False
    f16 = pair == 'f16'
    with x3_mode(ops, 'x3'):
        # fp16 pairs: the forward epilogues run on the large-M kernel, a BNRELU_BWD epilogue falls to gemm.hip's fp16-pair variant
        want = 0 if (f16 and case[4] == 2) else 1
        assert ops.gemm_path(**_with_images(ops, jd, f16)) == want, 'kernel selection (gemm_big.hip = 1)'
        ops.gemm(**jd)
    kc.gemm(**jh)
    torch.cuda.synchronize()
    _gemm_check(case, torch.float32, bufs, 'x3h' if f16 else 'x3')


@pytest.mark.parametrize('pair', PAIRS)
@pytest.mark.parametrize('nodrop', [False, True], ids=['xdrop', 'noxdrop'])
@pytest.mark.parametrize('first', [0, 3, 6])
def test_gemm_big_x3_multi(ops, first, nodrop, pair):
    """every epilogue variant of the large-M kernel also inside a multi-job grid (3 jobs per launch)"""
    cases = (GEMM_BIG_CASES + GEMM_BIG_CASES[:1])[first:first + 3]
    built = [_gemm_case(c, torch.float32) for c in cases]
    if nodrop:
        for jd, jh, _ in built:
            jd['xdrop'] = jh['xdrop'] = False
    f16 = pair == 'f16'
    with x3_mode(ops, 'x3'):
        ops.gemm_multi([_with_images(ops, jd, f16) for jd, _, _ in built])
    torch.cuda.synchronize()
    for c, (jd, jh, bufs) in zip(cases, built):
        kc.gemm(**jh)
        _gemm_check(c, torch.float32, bufs, 'x3h' if f16 else 'x3')


def _with_h16_images(ops, jd):
    for s in jd['segs']:
        s['W'] = ops.h16_weight(s['W'])
    return jd


@pytest.mark.parametrize('nodrop', [False, True], ids=['xdrop', 'noxdrop'])
@pytest.mark.parametrize('case', GEMM_BIG_CASES, ids=[c[0] for c in GEMM_BIG_CASES])
def test_gemm_big_bf16_storage(ops, case, nodrop):
    """Round 5: the large-M kernel on 16-bit STORAGE (GAST_BF16 tensors: bfloat16 here, binary16 when the suite runs in the f16 flavour),
    one product per value, 32 K values per step, weights from the k-group-major layout image (HipOps.h16_weight) -- every epilogue,
    row maps, prologue, K tails; same numpy contract as the 128 x 128 kernel's 16-bit cases."""
    if nodrop and case[4] != 2:
        pytest.skip('only the BNRELU_BWD epilogue has a dropout variant')
    jd, jh, bufs = _gemm_case(case, H16)
    if nodrop:
        jd['xdrop'] = jh['xdrop'] = False
    assert ops.gemm_path(**_with_h16_images(ops, jd)) == 1, 'kernel selection (gemm_big.hip = 1)'
    ops.gemm(**jd)
    kc.gemm(**jh)
    torch.cuda.synchronize()
    _gemm_check(case, H16, bufs, 'bf16')


@pytest.mark.parametrize('first', [0, 3, 6])
def test_gemm_big_bf16_storage_multi(ops, first):
    cases = (GEMM_BIG_CASES + GEMM_BIG_CASES[:1])[first:first + 3]
    built = [_gemm_case(c, H16) for c in cases]
    ops.gemm_multi([_with_h16_images(ops, jd) for jd, _, _ in built])
    torch.cuda.synchronize()
    for c, (jd, jh, bufs) in zip(cases, built):
        kc.gemm(**jh)
        _gemm_check(c, H16, bufs, 'bf16')


def test_gemm_big_bf16_storage_matches_the_small_kernel(ops):
    """the same 16-bit GEMM with and without the layout image: two kernels, two summation orders, one contract"""
    case = GEMM_BIG_CASES[1]
    jd, _, bufs = _gemm_case(case, H16)
    assert ops.gemm_path(**jd) == 0
    ops.gemm(**jd)
    torch.cuda.synchronize()
    C0, p0 = bufs[0].clone(), bufs[2].clone()
    bufs[0].fill_(7.0)
    assert ops.gemm_path(**_with_h16_images(ops, jd)) == 1
    ops.gemm(**jd)
    torch.cuda.synchronize()
    close(host(bufs[0]), host(C0), H16, 'C')
    close(host(bufs[2]).sum(axis=0), host(p0).sum(axis=0), H16, 'column sums', bf16=3e-2)


BWD_CASES = [c for c in GEMM_CASES + GEMM_BIG_CASES if c[4] == 2]


def _assert_bit_equal(got, ref, what):
    """torch.equal with a post-mortem: which elements differ, by how much, and whether a NaN (equal to nothing) is involved -- a bit
    miss on these outputs is a race or an uninitialised read (no atomics feed them), so the failure has to say WHERE."""
    g, r = got.detach().cpu(), ref.detach().cpu()
    same = (g.view(torch.int16 if g.element_size() == 2 else torch.int32) == r.view(torch.int16 if r.element_size() == 2 else torch.int32))
    if bool(same.all()):
        return
    bad = (~same).nonzero()
    rows, cols = bad[:, 0], bad[:, 1]
    gf, rf = g.float(), r.float()
    first = [(int(i), int(j), float(gf[i, j]), float(rf[i, j])) for i, j in bad[:12].tolist()]
    raise AssertionError('%s: %d of %d elements differ bitwise; rows %d..%d (%d distinct, 128-row tiles %s), columns %d..%d; NaN in got: %d, '
                         'in ref: %d; max |diff| %.3e; first (row, col, got, ref): %s'
                         % (what, bad.shape[0], g.numel(), int(rows.min()), int(rows.max()), int(rows.unique().numel()),
                            sorted(set((rows // 128).tolist()))[:16], int(cols.min()), int(cols.max()), int(torch.isnan(gf).sum()),
                            int(torch.isnan(rf).sum()), float(torch.nan_to_num(gf - rf).abs().max()), first))


@pytest.mark.parametrize('mode', ['f32', 'bf16', 'x3'])
@pytest.mark.parametrize('case', BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_gemm_bwd_second_output(ops, case, mode):
    """gast_gemm_args.C2: the BNRELU_BWD epilogue also stores the value BEFORE the mask (acc + bias + addend) -- bit-equal to the same
    GEMM with the PLAIN epilogue -- while C, the masked value, and the column sums are unchanged by it.  Both kernels, split-K finish."""
    dt = MM_DT[mode]
    jd, jh, bufs = _gemm_case(case, dt)
    big = case in GEMM_BIG_CASES
    with x3_mode(ops, mode):
        if big and mode == 'x3':
            _with_images(ops, jd)
        if big and mode == 'bf16':
            _with_h16_images(ops, jd)
        ops.gemm(**jd)
        torch.cuda.synchronize()
        C_ref, part_ref = bufs[0].clone(), bufs[2].clone()
        plain = torch.full_like(bufs[0], 7.0)
        jp = dict(jd, C_=plain[:, :case[2]], epi=0, partials=None, X=None, xscale=None, xshift=None, xdrop=False)
        ops.gemm(**jp)
        bufs[0].fill_(7.0)
        bufs[2].zero_()
        C2 = torch.full_like(bufs[0], 7.0)
        ops.gemm(**dict(jd, C2=C2[:, :case[2]]))
        torch.cuda.synchronize()
    if big and mode in ('x3', 'bf16') and jd.get('addend') is not None:
        # (the large-M kernel carries the second output in its addend-free variants only: with an addend this call takes the 128x128-tile
        #  kernel -- another summation order than the large-M run it is compared with)
        close(host(bufs[0]), host(C_ref), dt, 'masked output', fp32=1e-4)
        close(host(C2), host(plain), dt, 'C2', fp32=1e-4)
    else:
        _assert_bit_equal(bufs[0], C_ref, 'the masked output (C) of the call WITH a second output vs the call without')
        _assert_bit_equal(C2, plain, 'C2 vs the PLAIN epilogue of the same GEMM')
    # (16-bit storage, two kernels: a value that rounds the other way in ONE of the two summation orders moves a column sum by an ulp of
    #  the stored type -- the 128 x 128 kernel's own 16-bit bound applies there, not the same-kernel one)
    two_kernels = big and mode == 'bf16' and jd.get('addend') is not None
    close(host(bufs[2]).sum(axis=0), host(part_ref).sum(axis=0), dt, 'sums', fp32=1e-5, bf16=3e-2 if two_kernels else 1e-5)


def test_gemm_fp16_pairs_range(ops):
    """The documented range of GAST_F32X3H (include/gast_hip.h): operands up to fp16's largest finite value are exact-class, parts
    below 2^-24 vanish (absolute, not relative, error floor), and an operand beyond 65504 is NOT representable -- the result is
    non-finite, never silently wrong; the same inputs on bf16 pairs (GAST_F32X3, fp32's exponent range) stay fp32-class.  The forward
    GEMMs of the plan read post-BatchNorm activations, their aggregates and weights: O(1e-3 .. 1e2)."""
    from gast_hip.packer import X3Weight
    gen = torch.Generator().manual_seed(5)
    M, K, N = 17 * 8, 64, 32
    A = torch.rand(M, K, generator=gen) + 0.5
    W = (torch.rand(N, K, generator=gen) + 0.5) / K
    dom, im = (8, 1, 17), kc.RowMap(1, 1, 0)

    def run(scale_a, f16):
        Cd = torch.empty(M, N).cuda()
        with x3_mode(ops, 'x3'):
            ops.gemm(dom, N, [dict(A=(A * scale_a).cuda(), K=K, map=im, W=X3Weight(W.cuda(), None, f16))], Cd, im)
        torch.cuda.synchronize()
        return Cd.cpu().double(), (A.double() * scale_a) @ W.double().t()
    # (3e4 * 1.5 = 4.5e4 < 65504; at 1e-2 the lo halves (~5e-6) already sit in fp16's subnormals: absolute floor 3e-8 per element)
    for scale, tol in ((1.0, 2e-6), (3.0e4, 2e-6), (1.0e-2, 1e-5)):
        got, ref = run(scale, True)
        assert float((got - ref).abs().max() / ref.abs().max()) < tol, scale
    got, ref = run(1.0e-7, True)                              # lo parts (and most hi bits) below fp16's subnormals: ABSOLUTE floor ~6e-8 * |w| * K
    assert float((got - ref).abs().max()) < 1e-7 and float((got - ref).abs().max() / ref.abs().max()) > 1e-4
    got, _ = run(1.0e5, True)                                 # beyond fp16: loudly non-finite
    assert not torch.isfinite(got).all()
    for scale in (1.0e5, 1.0e-7, 1.0e20):                     # bf16 pairs keep fp32's range
        got, ref = run(scale, False)
        assert torch.isfinite(got).all() and float((got - ref).abs().max() / ref.abs().max()) < 1e-4, scale


def test_gemm_pairs_do_not_mix(ops):
    """one GEMM, one kind of operand pairs: fp16-tagged and untagged weight segments in one call are refused (binding), jobs of
    different kinds in one multi call as well (C ABI: dtypes differ)"""
    jd, _, _ = _gemm_case(GEMM_CASES[2], torch.float32)
    from gast_hip.packer import X3Weight
    jd['segs'][0]['W'] = X3Weight(jd['segs'][0]['W'], None, True)
    with x3_mode(ops, 'x3'):
        with pytest.raises(RuntimeError, match='fp16-pair and bf16-pair'):
            ops.gemm(**jd)
        a, _, _ = _gemm_case(GEMM_CASES[0], torch.float32)
        b, _, _ = _gemm_case(GEMM_CASES[1], torch.float32)
        with pytest.raises(RuntimeError):
            ops.gemm_multi([_f16_pairs(a), b])


GEMM_SMALL_X3_CASES = [
    ('small_stats_k1536', (128, 1, 17), 512, [(512, 3, 1, 0, 1), (512, 3, 1, 1, 1), (512, 3, 1, 2, 1)], 1, False, False),
    ('small_plain_wide', (128, 1, 17), 648, [(256, 1, 1, 0, 1)], 0, False, True),
    ('small_bwd_add', (96, 1, 19), 256, [(520, 1, 1, 0, 0)], 2, True, False),
]


@pytest.mark.parametrize('pair', PAIRS)
@pytest.mark.parametrize('case', GEMM_SMALL_X3_CASES, ids=[c[0] for c in GEMM_SMALL_X3_CASES])
def test_gemm_small_x3_images(ops, case, pair):
    """The M = B*J stage with pre-split weight images attached to its operands: gemm.hip's split-K path reads the fp32 weights."""
    jd, jh, bufs = _gemm_case(case, torch.float32)
    with x3_mode(ops, 'x3'):
        ops.gemm(**_with_images(ops, jd, pair == 'f16'))
    kc.gemm(**jh)
    torch.cuda.synchronize()
    _gemm_check(case, torch.float32, bufs, 'x3h' if pair == 'f16' else 'x3')


# the M = B*J kernel (csrc/gemm_bj.hip): every small-M case with weight images attached runs on it unless a segment carries the dropout
# prologue (gemm.hip keeps that variant) -- row maps, K tails (168 = 5*32 + 8: a second k-group past the segment's end; 2568 = 160*16 + 8),
# one-step K = 8, N = 3 (the shrink layer), both epilogue families, addend, centred bias, N tails
GEMM_BJ_CASES = GEMM_CASES + GEMM_SMALL_X3_CASES + [
    ('bj_g1_bwd_k2568', (40, 1, 17), 136, [(2568, 1, 1, 0, 0), (1024, 1, 1, 0, 0)], 2, False, False),
    ('bj_k8_bwd', (128, 1, 17), 1024, [(8, 1, 1, 0, 0)], 2, False, False),
    ('bj_shrink_n3', (128, 1, 17), 3, [(1024, 1, 1, 0, 1)], 0, False, False),
    ('bj_scatter_taps_add', (50, 1, 17), 192, [(256, 1, 1, 0, 0)], 2, True, False),
    ('bj_wide_n2568', (33, 1, 17), 2568, [(128, 1, 1, 0, 0)], 0, False, True),
    ('bj_two_pro_tables', (20, 3, 15), 100, [(200, 3, 1, 0, 1), (72, 5, 1, 2, 1), (40, 3, 1, 0, 0)], 1, True, 'neg'),
    # the lean K loop (every K a multiple of 128, rows a multiple of 64, full row maps): prologue / plain segments mixed, taps, both epilogues
    ('bj_fast_pro_mix', (64, 2, 17), 96, [(128, 4, 1, 1, 1), (384, 2, 1, 0, 0), (128, 4, 2, 0, 1)], 1, False, 'neg'),
    ('bj_fast_concat_bwd_add', (128, 1, 17), 192, [(256, 1, 1, 0, 0), (128, 3, 1, 2, 0)], 2, True, False),
    ('bj_fast_one_group', (64, 1, 17), 64, [(128, 1, 1, 0, 1)], 1, False, True),
    # ... and a ragged last row tile (M = 119 and 2 159: rows past M are clamped on load, never stored, never counted)
    ('bj_fast_ragged_m119', (7, 1, 17), 128, [(256, 1, 1, 0, 1)], 1, False, True),
    ('bj_fast_ragged_bwd_add', (127, 1, 17), 192, [(128, 1, 1, 0, 0), (128, 3, 1, 2, 0)], 2, True, False),
]


def _bj_expected(case, f16):
    if any(sd[4] == 2 for sd in case[3]) or (f16 and case[4] == 2):
        return 0
    return 2


@pytest.mark.parametrize('pair', PAIRS)
@pytest.mark.parametrize('nodrop', [False, True], ids=['xdrop', 'noxdrop'])
@pytest.mark.parametrize('case', GEMM_BJ_CASES, ids=[c[0] for c in GEMM_BJ_CASES])
def test_gemm_bj_x3(ops, case, nodrop, pair, monkeypatch):
    if nodrop and case[4] != 2:
        pytest.skip('only the BNRELU_BWD epilogue has a dropout variant')
    f16 = pair == 'f16'
    outs = []
    for rep in range(2):
        jd, jh, bufs = _gemm_case(case, torch.float32)
        if nodrop:
            jd['xdrop'] = jh['xdrop'] = False
        with x3_mode(ops, 'x3'):
            assert ops.gemm_path(**_with_images(ops, jd, f16)) == _bj_expected(case, f16), 'kernel selection (gemm_bj.hip = 2)'
            ops.gemm(**jd)
        torch.cuda.synchronize()
        outs.append(bufs)
    kc.gemm(**jh)
    _gemm_check(case, torch.float32, outs[1], 'x3h' if f16 else 'x3')
    if _bj_expected(case, f16) == 2:      # no cross-block reduction on the output, two commutative adds per statistics element
        _assert_bit_equal(outs[1][0], outs[0][0], case[0] + ': output of two runs')
        if case[4]:
            _assert_bit_equal(outs[1][2].view(outs[1][2].shape[0], -1), outs[0][2].view(outs[0][2].shape[0], -1), case[0] + ': column statistics of two runs')


@pytest.mark.parametrize('pair', PAIRS)
@pytest.mark.parametrize('first', list(range(0, len(GEMM_BJ_CASES), 3)))
def test_gemm_bj_x3_multi(ops, first, pair):
    """the same cases as jobs of multi-job launches (3 per call; a call may mix the M = B*J kernel with gemm.hip's for a dropout prologue)"""
    cases = GEMM_BJ_CASES[first:first + 3]
    built = [_gemm_case(c, torch.float32) for c in cases]
    f16 = pair == 'f16'
    with x3_mode(ops, 'x3'):
        ops.gemm_multi([_with_images(ops, jd, f16) for jd, _, _ in built])
    torch.cuda.synchronize()
    for c, (jd, jh, bufs) in zip(cases, built):
        kc.gemm(**jh)
        _gemm_check(c, torch.float32, bufs, 'x3h' if f16 else 'x3')


# Round 6: the M = B*J kernel's lean loop on 16-bit STORAGE (PAIR = 3): every K a multiple of 256 (a group of four 64-value steps), full
# row maps, any row count; weights from the layout image
GEMM_BJ_H16_CASES = [
    ('bjh_pro_mix', (64, 2, 17), 96, [(256, 4, 1, 1, 1), (512, 2, 1, 0, 0), (256, 4, 2, 0, 1)], 1, False, 'neg'),
    ('bjh_concat_bwd_add', (128, 1, 17), 192, [(256, 1, 1, 0, 0), (256, 3, 1, 2, 0)], 2, True, False),
    ('bjh_one_group', (64, 1, 17), 64, [(256, 1, 1, 0, 1)], 1, False, True),
    ('bjh_plain_n512', (128, 1, 17), 512, [(1024, 1, 1, 0, 1)], 0, False, False),
    ('bjh_bwd_n1024', (128, 1, 17), 1024, [(512, 1, 1, 0, 0)], 2, False, False),
    ('bjh_ragged_m2159', (127, 1, 17), 192, [(256, 1, 1, 0, 0), (256, 3, 1, 2, 1)], 2, True, False),
    ('bjh_ragged_m45', (3, 1, 15), 64, [(256, 1, 1, 0, 1)], 1, False, 'neg'),
]


@pytest.mark.parametrize('nodrop', [False, True], ids=['xdrop', 'noxdrop'])
@pytest.mark.parametrize('case', GEMM_BJ_H16_CASES, ids=[c[0] for c in GEMM_BJ_H16_CASES])
def test_gemm_bj_bf16_storage(ops, case, nodrop):
    """gemm_bj.hip on GAST_BF16 tensors (bfloat16 here, binary16 when the suite runs in the f16 flavour): one product per value, 64 K
    values per step, every epilogue; same numpy contract as gemm.hip's 16-bit cases, and run-to-run bit-equal (no cross-block reduction
    on the output, two commutative adds per statistics element)."""
    if nodrop and case[4] != 2:
        pytest.skip('only the BNRELU_BWD epilogue has a dropout variant')
    if _os.environ.get('GAST_GEMM_BJ_FAST') == '0':
        pytest.skip('16-bit storage runs on the lean loop only (the opt-in child run switched it off)')
    outs = []
    for rep in range(2):
        jd, jh, bufs = _gemm_case(case, H16)
        if nodrop:
            jd['xdrop'] = jh['xdrop'] = False
        assert ops.gemm_path(**_with_h16_images(ops, jd)) == 2, 'kernel selection (gemm_bj.hip = 2)'
        ops.gemm(**jd)
        torch.cuda.synchronize()
        outs.append(bufs)
    kc.gemm(**jh)
    _gemm_check(case, H16, outs[1], 'bf16')
    _assert_bit_equal(outs[1][0], outs[0][0], case[0] + ': output of two runs')
    if case[4]:
        _assert_bit_equal(outs[1][2].view(outs[1][2].shape[0], -1), outs[0][2].view(outs[0][2].shape[0], -1), case[0] + ': column statistics of two runs')


def test_gemm_bj_bf16_storage_multi(ops):
    if _os.environ.get('GAST_GEMM_BJ_FAST') == '0':
        pytest.skip('16-bit storage runs on the lean loop only')
    cases = GEMM_BJ_H16_CASES[:3]
    built = [_gemm_case(c, H16) for c in cases]
    ops.gemm_multi([_with_h16_images(ops, jd) for jd, _, _ in built])
    torch.cuda.synchronize()
    for c, (jd, jh, bufs) in zip(cases, built):
        kc.gemm(**jh)
        _gemm_check(c, H16, bufs, 'bf16')


def test_gemm_bj_bf16_storage_keeps_irregular_shapes_on_the_small_kernel(ops):
    """a K that is no multiple of 256, or a row map that leaves the tensor, stays on gemm.hip in 16-bit storage (the lean loop is the only one)"""
    for case in [('k384', (128, 1, 17), 192, [(384, 1, 1, 0, 0)], 1, False, False), ('tap_out_of_range', (64, 2, 17), 192, [(256, 2, 1, 1, 0)], 1, False, False)]:
        jd, _, _ = _gemm_case(case, H16)
        assert ops.gemm_path(**_with_h16_images(ops, jd)) == 0, case[0]


@pytest.mark.parametrize('case', [c for c in GEMM_BJ_CASES if c[4] == 2], ids=lambda c: c[0])
def test_gemm_bj_second_output(ops, case):
    """gast_gemm_args.C2 on the M = B*J kernel, with and without an addend: bit-equal to the PLAIN epilogue of the same GEMM; C and
    the column sums unchanged by it"""
    jd, jh, bufs = _gemm_case(case, torch.float32)
    N = case[2]
    with x3_mode(ops, 'x3'):
        _with_images(ops, jd)
        assert ops.gemm_path(**jd) == 2
        ops.gemm(**jd)
        torch.cuda.synchronize()
        C_ref, part_ref = bufs[0].clone(), bufs[2].clone()
        plain = torch.full_like(bufs[0], 7.0)
        ops.gemm(**dict(jd, C_=plain[:, :N], epi=0, partials=None, X=None, xscale=None, xshift=None, xdrop=False))
        bufs[0].fill_(7.0)
        bufs[2].zero_()
        C2 = torch.full_like(bufs[0], 7.0)
        ops.gemm(**dict(jd, C2=C2[:, :N]))
        torch.cuda.synchronize()
    _assert_bit_equal(bufs[0], C_ref, 'the masked output (C) of the call WITH a second output vs the call without')
    _assert_bit_equal(C2, plain, 'C2 vs the PLAIN epilogue of the same GEMM')
    _assert_bit_equal(bufs[2].view(bufs[2].shape[0], -1), part_ref.view(part_ref.shape[0], -1), 'column sums')


GUARD_CASES = [
    # M = 1 (mod 128) rows, every K = 16 (mod 32): the last row tile holds one row, every segment ends in half a K step
    ('guard_small_m', (1, 43, 3), 72, [(48, 43, 1, 0, 1), (80, 45, 1, 2, 0)], 1, False, True),              # M = 129
    ('guard_bwd_m', (5, 7, 11), 136, [(112, 7, 1, 0, 0)], 2, True, False),                                    # M = 385
    ('guard_big_m', (1, 2731, 3), 96, [(48, 2731, 1, 0, 1), (176, 2735, 1, 4, 0)], 1, False, 'neg'),         # M = 8193: the large-M kernel
]


@pytest.mark.parametrize('images', [False, True], ids=['fp32_operands', 'images'])
@pytest.mark.parametrize('case', GUARD_CASES, ids=[c[0] for c in GUARD_CASES])
def test_gemm_guard_bands(ops, case, images):
    """ragged shapes on every GEMM kernel (gemm.hip without images, gemm_bj.hip / gemm_big.hip with) with the OUTPUT,
    the second output and the column statistics each carved out of a NaN-filled arena -- 4 KB guard bands on both sides must come back
    untouched -- and every activation / weight operand ending exactly at the end of its allocation (an over-read of the last row or
    the last K step leaves the tensor).  Results against the contract."""
    name, dom, N, segdefs, epi, use_add, use_bias = case
    B, Tn, J = dom
    M = B * Tn * J
    assert M % 128 == 1 and all(sd[0] % 32 == 16 for sd in segdefs)
    jd, jh, bufs = _gemm_case(case, torch.float32)
    Cd, Ch, pd, ph = bufs
    G = 1024                                             # guard band: 1024 floats
    rowsC, ldc = Cd.shape
    sizes = [rowsC * ldc, rowsC * ldc if epi == 2 else 0, pd.numel() if pd is not None else 0]
    arena = torch.full((sum(sizes) + G * (len(sizes) + 1),), float('nan')).cuda()
    offs, o = [], G
    for n in sizes:
        offs.append(o)
        o += n + G
    Cg = arena[offs[0]:offs[0] + sizes[0]].view(rowsC, ldc)
    Cg.fill_(7.0)
    jd['C_'] = Cg[:, :N]
    C2g = None
    if epi == 2:
        C2g = arena[offs[1]:offs[1] + sizes[1]].view(rowsC, ldc)
        C2g.fill_(7.0)
        jd['C2'] = C2g[:, :N]
    if pd is not None:
        pg = arena[offs[2]:offs[2] + sizes[2]].view(pd.shape)
        pg.zero_()
        jd['partials'] = pg
    with x3_mode(ops, 'x3'):
        if images:
            _with_images(ops, jd)
            assert ops.gemm_path(**jd) == (1 if M >= 8192 else 2)
        ops.gemm(**jd)
    torch.cuda.synchronize()
    kc.gemm(**jh)
    a = arena.cpu()
    inside = torch.zeros(a.numel(), dtype=torch.bool)
    for of, n in zip(offs, sizes):
        inside[of:of + n] = True
    assert bool(torch.isnan(a[~inside]).all()), 'a guard band was written: elements %s' % (~inside & ~torch.isnan(a)).nonzero()[:8].flatten().tolist()
    got = Cg.cpu().numpy()
    close(got[:, :N], Ch[:, :N], torch.float32, name + ' C', fp32=1e-4)
    assert np.all(got[:, N:] == 7.0), 'wrote outside the N columns'
    if C2g is not None:
        assert np.all(C2g.cpu().numpy()[:, N:] == 7.0)
    if pd is not None:
        close(host(jd['partials']).sum(axis=0), ph.sum(axis=0), torch.float32, name + ' partial totals', fp32=2e-4)


def test_x3_image_layout(ops):
    """k-group-major image: img[k>>4][r][k&15] = bf16(w), [...][16 + (k&15)] = bf16(w - hi); zero K padding and zero rows behind"""
    gen = torch.Generator().manual_seed(11)
    W = rand(gen, 37, 72).cuda()
    xw = ops.x3_weight(W)
    torch.cuda.synchronize()
    img = xw.img.float().cpu().numpy()                # [5][48 + 256][32]
    assert img.shape == (5, 48 + 256, 32)
    Wp = np.zeros((37, 80), dtype=np.float32)
    Wp[:, :72] = W.cpu().numpy()
    hi = torch.from_numpy(Wp).to(torch.bfloat16).float().numpy()
    lo = torch.from_numpy(Wp - hi).to(torch.bfloat16).float().numpy()
    assert np.array_equal(img[:, :37, :16].transpose(1, 0, 2).reshape(37, 80), hi)
    assert np.array_equal(img[:, :37, 16:].transpose(1, 0, 2).reshape(37, 80), lo)
    assert not img[:, 37:].any()
    sl = xw[5:9, 32:72]                  # aligned column slice keeps its image, an unaligned one drops it
    assert sl.img is not None and sl.img.data_ptr() == xw.img[2:, 5:].data_ptr() and xw[:, 8:40].img is None
    # the fp16 kind (GAST_F32X3H): same layout, fp16 halves (subnormal lo parts included); slices keep the tag
    Ws = W * torch.logspace(-4, 0, 72, device='cuda')            # columns down to 1e-4: lo parts reach fp16's subnormals
    xh = ops.x3_weight(Ws, True)
    torch.cuda.synchronize()
    imh = xh.img.view(torch.float16).float().cpu().numpy()
    Wp[:, :72] = Ws.cpu().numpy()
    hi = torch.from_numpy(Wp).to(torch.float16).float().numpy()
    lo = torch.from_numpy(Wp - hi).to(torch.float16).float().numpy()
    assert np.array_equal(imh[:, :37, :16].transpose(1, 0, 2).reshape(37, 80), hi)
    assert np.array_equal(imh[:, :37, 16:].transpose(1, 0, 2).reshape(37, 80), lo)
    assert (np.abs(lo[lo != 0]) < 6.1e-5).any(), 'the case is meant to exercise subnormal lo parts'
    assert not imh[:, 37:].any() and xh.f16 and xh[5:9, 32:72].f16 and not xw.f16


@pytest.mark.parametrize('case', [c for c in GEMM_CASES if c[0] in ('stats_two_tiles', 'dilated_taps', 'ktail', 'big', 'splitk_stats', 'concat3_drop')],
                         ids=lambda c: c[0])
def test_gemm_fp8_operands(ops, case):
    """gast_gemm_args.f8_scale ("mixed fp8", BASELINE.json configs[4]): bf16 storage, operands as OCP e4m3 on the fp8 matrix
    instruction -- against the contract evaluated on e4m3-rounded operands (which pins the number format: the FNUZ format of the
    previous GPU generation would be off by a factor of two)."""
    from gast_hip.packer import F8Weight
    jd, jh, bufs = _gemm_case(case, torch.bfloat16)
    # one weight scale per GEMM: computed by the library's kernel from the first segment's weights, checked against the contract
    Wall = torch.cat([s['W'].reshape(-1) for s in jd['segs']]).view(1, -1).contiguous()
    sc = torch.ones(1, 2, device='cuda')
    from gast_hip.binding import _F8ScaleJob, _p
    import ctypes
    job = (_F8ScaleJob * 1)()
    job[0].W, job[0].R, job[0].K, job[0].ldw, job[0].out = _p(Wall), 1, Wall.shape[1], Wall.shape[1], _p(sc)
    assert ops.lib.gast_f8_scale_multi(job, 1, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    s_ref = kc.f8_weight_scale(np.concatenate([h['W'].reshape(-1) for h in jh['segs']]))
    assert float(sc[0, 0]) == s_ref and float(sc[0, 1]) == 1.0 / s_ref
    for s in jd['segs']:
        s['W'] = F8Weight(s['W'], sc[0])
    ops.f8 = True
    try:
        ops.gemm(**jd)
    finally:
        ops.f8 = False
    kc.gemm(f8_scale=s_ref, **jh)
    torch.cuda.synchronize()
    _gemm_check(case, torch.bfloat16, bufs)


def test_gemm_out_f32_from_bf16(ops):
    gen = torch.Generator().manual_seed(5)
    dom, N, K = (2, 4, 17), 3, 64
    M = 2 * 4 * 17
    A = rand(gen, M, K).to(H16)
    W = rand(gen, N, K).to(H16)
    C = torch.zeros(M, N).cuda()
    ops.gemm(dom, N, [dict(A=A.cuda(), K=K, map=kc.RowMap(4, 1, 0), W=W.cuda())], C, kc.RowMap(4, 1, 0))
    ref = host(A) @ host(W).T
    close(host(C), ref, torch.float32, 'bf16->f32 gemm', fp32=1e-5)


# ------------------------------------------------------------------------------------------------ WGRAD
WGRAD_CASES = [
    ('one_seg', (2, 5, 17), 48, [(64, 5, 1, 0, 0, 0)]),
    ('concat3_drop', (3, 7, 17), 64, [(32, 7, 1, 0, 0, 0), (32, 7, 1, 0, 2, 32), (32, 7, 1, 0, 2, 64)]),
    ('dilated_taps', (2, 5, 17), 32, [(32, 11, 1, 0, 1, 0), (32, 11, 1, 3, 1, 32), (32, 11, 1, 6, 1, 64)]),
    ('strided_taps', (4, 3, 15), 24, [(24, 9, 3, 0, 1, 0), (24, 9, 3, 1, 1, 24), (24, 9, 3, 2, 1, 48)]),
    ('big', (8, 27, 17), 136, [(160, 27, 1, 0, 1, 0)]),
    ('pad8', (2, 1, 17), 8, [(128, 1, 1, 0, 1, 0)]),
]


def _wgrad_case(case, dt):
    """device job (keyword arguments of ops.wgrad), host job (for kc.wgrad) and the two dW buffers of one WGRAD_CASES entry"""
    from gast_hip.binding import Dropout, dropout_params
    name, dom, R, segdefs = case
    gen = torch.Generator().manual_seed(sum(map(ord, name)) + 1)
    B, Tn, J = dom
    M = B * Tn * J
    thresh, inv_keep = dropout_params(0.1)
    P = rand(gen, M, R + 8).to(dt)
    segs_d, segs_h = [], []
    ldw = max(w0 + S for (S, _, _, _, _, w0) in segdefs) + 4
    for si, (S, Tt, ts, toff, pro, w0) in enumerate(segdefs):
        rowsQ = B * Tt * J
        Q = rand(gen, rowsQ, S + 8).to(dt)
        sc = torch.rand(S, generator=gen) + 0.5
        sh = rand(gen, S, scale=0.3)
        segs_d.append(dict(Q=Q.cuda()[:, :S], S=S, map=kc.RowMap(Tt, ts, toff), pro=pro, scale=sc.cuda(), shift=sh.cuda(), salt=si + 3, wcol0=w0))
        hQ = host(Q)
        segs_h.append(dict(Q=hQ[:, :S], S=S, map=kc.RowMap(Tt, ts, toff), pro=pro, scale=host(sc), shift=host(sh), salt=si + 3, wcol0=w0))
    dWd = torch.full((R, ldw), 3.0).cuda()
    dWh = np.full((R, ldw), 3.0)
    jd = dict(dom=dom, P=P.cuda()[:, :R], R=R, pmap=kc.RowMap(Tn, 1, 0), segs=segs_d, dW=dWd, drop=Dropout(seed_tensor(99), thresh, inv_keep))
    jh = dict(dom=dom, P=host(P)[:, :R], R=R, pmap=kc.RowMap(Tn, 1, 0), segs=segs_h, dW=dWh, drop=(99, thresh, inv_keep))
    return jd, jh


@pytest.mark.parametrize('mode', MM_MODES[:3])      # (weight gradients: bf16 pairs only -- gradient operands need fp32's range)
@pytest.mark.parametrize('case', WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_wgrad(ops, case, mode):
    dt = MM_DT[mode]
    jd, jh = _wgrad_case(case, dt)
    with x3_mode(ops, mode):
        ops.wgrad(**jd)
    kc.wgrad(jh['dom'], jh['P'], jh['R'], jh['pmap'], jh['segs'], jh['dW'], drop=jh['drop'])
    torch.cuda.synchronize()
    close(host(jd['dW']), jh['dW'], dt, case[0], fp32=1e-4 if mode == 'x3' else 3e-5, bf16=2e-2)


@pytest.mark.parametrize('mode', MM_MODES[:3])      # (weight gradients: bf16 pairs only -- gradient operands need fp32's range)
def test_wgrad_multi(ops, mode):
    """All WGRAD_CASES (different domains, segment counts, prologues) as ONE multi-job launch, accumulating into the 3.0 fill
    (zero_first=False) for the odd jobs and overwriting it for the even ones."""
    dt = MM_DT[mode]
    jobs = [_wgrad_case(c, dt) for c in WGRAD_CASES]
    for i, (jd, jh) in enumerate(jobs):
        jd['zero_first'] = i % 2 == 0
    with x3_mode(ops, mode):
        ops.wgrad_multi([jd for jd, _ in jobs])
    torch.cuda.synchronize()
    for i, (jd, jh) in enumerate(jobs):
        kc.wgrad(jh['dom'], jh['P'], jh['R'], jh['pmap'], jh['segs'], jh['dW'], jh['drop'], i % 2 == 0)
        close(host(jd['dW']), jh['dW'], dt, 'multi ' + WGRAD_CASES[i][0], fp32=1e-4 if mode == 'x3' else 3e-5, bf16=2e-2)


WGRAD_WIDE_CASES = [      # job sets that fill 256 x 256 tiles (the library then picks wgrad_wide.hip): edge tiles, taps, prologue, dropout, ragged chunk
    ('wide_taps_drop', (6, 9, 17), 512, [(256, 11, 1, 0, 1, 0), (256, 11, 1, 2, 2, 256)]),
    ('wide_edge', (6, 9, 17), 488, [(256, 9, 1, 0, 0, 0)]),
    ('wide_concat', (6, 9, 17), 256, [(512, 9, 1, 0, 1, 0), (256, 9, 1, 0, 2, 512)]),
    ('wide_strided_taps', (4, 3, 19), 256, [(256, 9, 3, 0, 1, 0), (256, 9, 3, 1, 0, 256), (256, 9, 3, 2, 1, 512)]),
]


def test_wgrad_multi_wide_tiles(ops):
    """bf16x3 weight gradients of matrices that fill 256 x 256 tiles: one launch of wgrad_wide.hip (8 waves, three tiles of operand
    rows in flight) -- several output tiles per job, an edge tile (R = 488), dilated taps, the BN+ReLU and dropout prologues, and 918
    reduction rows in chunks of 96 (a ragged last chunk).  GAST_WGRAD_X3_TILE=128 (test_optin_kernel_variants) runs the same jobs on
    the 128 x 128 kernel."""
    jobs = [_wgrad_case(c, torch.float32) for c in WGRAD_WIDE_CASES]
    for i, (jd, jh) in enumerate(jobs):
        jd['zero_first'] = i % 2 == 0
    with x3_mode(ops, 'x3'):
        ops.wgrad_multi([jd for jd, _ in jobs])
    torch.cuda.synchronize()
    for i, (jd, jh) in enumerate(jobs):
        kc.wgrad(jh['dom'], jh['P'], jh['R'], jh['pmap'], jh['segs'], jh['dW'], jh['drop'], i % 2 == 0)
        close(host(jd['dW']), jh['dW'], torch.float32, 'wide ' + WGRAD_WIDE_CASES[i][0], fp32=1e-4)


def test_wide_wgrad_bf16_storage(ops):
    """Round 5: the same job sets on 16-bit STORAGE (bfloat16 / binary16 by flavour) -- wgrad_wide.hip in its one-product form, 32
    reduction rows per step (918 rows in chunks of 192: a ragged last chunk), v_perm transposition on the operand without a prologue.
    GAST_WGRAD_H16_WIDE=0 (test_optin_kernel_variants) runs them on the 128 x 128 kernel."""
    jobs = [_wgrad_case(c, H16) for c in WGRAD_WIDE_CASES]
    for i, (jd, jh) in enumerate(jobs):
        jd['zero_first'] = i % 2 == 0
    ops.wgrad_multi([jd for jd, _ in jobs])
    torch.cuda.synchronize()
    for i, (jd, jh) in enumerate(jobs):
        kc.wgrad(jh['dom'], jh['P'], jh['R'], jh['pmap'], jh['segs'], jh['dW'], jh['drop'], i % 2 == 0)
        close(host(jd['dW']), jh['dW'], H16, 'wide 16-bit ' + WGRAD_WIDE_CASES[i][0], bf16=2e-2)


@pytest.mark.parametrize('knob,select,npass', [('GAST_WGRAD_TILE=256', 'test_wgrad_multi and bf16', 1),
                                               ('GAST_WGRAD_H16_WIDE=0', 'test_wide_wgrad_bf16_storage', 1),
                                               ('GAST_WGRAD_X3_TILE=256', 'test_wgrad_multi and x3', 1),
                                               ('GAST_WGRAD_X3_TILE=128', 'test_wgrad_multi_wide_tiles', 1),
                                               ('GAST_WGRAD_RING=2', 'test_wgrad_multi and bf16', 1),
                                               ('GAST_WGRAD_ORDER=1', 'test_wgrad_multi and bf16', 1),
                                               ('GAST_WGRAD_X3_PIPE=0', 'test_wgrad_multi and x3', 1),
                                               ('GAST_AGG_BWD_LDS=0', 'semch_agg', None),
                                               ('GAST_GEMM_BIG_NI=2', 'test_gemm_big_x3', None),
                                               ('GAST_GEMM_BIG_NI=4', 'test_gemm_big_x3', None),
                                               ('GAST_GEMM_BIG_MW=4', 'test_gemm_big_x3', None),
                                               ('GAST_GEMM_BJ_OCC3_BLOCKS=0', 'test_gemm_bj', None),      # the 80-register lean loop (three blocks per CU) on every regular shape
                                               ('GAST_GEMM_BJ_FAST=0', 'test_gemm_bj', None),             # the general loop on every shape
                                               ('GAST_ATTN_MFMA=0', 'test_attention and bf16', None)])
def test_optin_kernel_variants(knob, select, npass):
    """Kernel variants behind environment switches (read once per process by the library): 256x256 weight-gradient tiles (forced on
    the small cases / forced off on the wide ones for the bf16x3 kernels), two
    register sets in flight, chunk-major block order, the un-pipelined bf16x3 weight-gradient kernel, the aggregation backward without LDS staging, both tile widths of the large-M GEMM on every case, and the VALU (non-MFMA) bf16 attention kernels -- the same parity cases
    in a child process."""
    import os
    import subprocess
    import sys
    k, v = knob.split('=')
    env = dict(os.environ, **{k: v})
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                        '-k', '(%s) and not optin' % select], env=env, capture_output=True, text=True, timeout=600)
    ok = r.returncode == 0 and (' passed' in r.stdout) and ('%d passed' % npass in r.stdout if npass else True)
    assert ok, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ SemCH
@pytest.mark.parametrize('J', [15, 16, 17, 19])
def test_semch_adj(ops, J):
    gen = torch.Generator().manual_seed(J)
    ps, pc = patterns(J)
    C = 24
    for pat in (ps, pc):
        nnz = int(pat[1])
        e = 1 + rand(gen, C, nnz, scale=0.5)
        A = torch.full((nnz + 1, C), 7.0).cuda()
        ops.semch_adj_fwd(e.cuda(), dev(pat), A)
        Ah = np.full((nnz + 1, C), 7.0)
        kc.semch_adj_fwd(host(e), pat, Ah)
        close(host(A), Ah, torch.float32, 'adj fwd')
        assert float(A[nnz].abs().max()) == 0.0
        dA = rand(gen, nnz, C)
        de = torch.zeros(C, nnz).cuda()
        ops.semch_adj_bwd(dA.cuda(), A, dev(pat), de)
        deh = np.zeros((C, nnz))
        kc.semch_adj_bwd(host(dA), Ah[:nnz], pat, deh)
        close(host(de), deh, torch.float32, 'adj bwd')
    # the multi-job launches: both graphs at three channel widths in one grid, forward then backward
    fj, bj, refs = [], [], []
    for C in (8, 40, 136):
        for pat in (ps, pc):
            nnz = int(pat[1])
            e = 1 + rand(gen, C, nnz, scale=0.5)
            A = torch.full((nnz + 1, C), 7.0).cuda()
            dA = rand(gen, nnz, C)
            de = torch.zeros(C, nnz).cuda()
            fj.append((e.cuda(), dev(pat), A))
            bj.append((dA.cuda(), A, dev(pat), de))
            Ah, deh = np.full((nnz + 1, C), 7.0), np.zeros((C, nnz))
            kc.semch_adj_fwd(host(e), pat, Ah)
            kc.semch_adj_bwd(host(dA), Ah[:nnz], pat, deh)
            refs.append((Ah, deh))
    ops.semch_adj_fwd_multi(fj)
    ops.semch_adj_bwd_multi(bj)
    for (_, _, A), (_, _, _, de), (Ah, deh) in zip(fj, bj, refs):
        close(host(A), Ah, torch.float32, 'adj fwd multi')
        close(host(de), deh, torch.float32, 'adj bwd multi')


@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('J,C,F', [(17, 16, 37), (19, 128, 50), (15, 8, 9), (16, 1024 + 64, 3), (17, 256, 700)])
def test_semch_agg(ops, J, C, F, dt):
    gen = torch.Generator().manual_seed(J * C)
    ps, pc = patterns(J)
    P = F * J
    ldh = 5 * C + 8
    H = rand(gen, P, ldh).to(dt)
    As, Ac = torch.rand(int(ps[1]) + 1, C, generator=gen), torch.rand(int(pc[1]) + 1, C, generator=gen)
    As[-1] = 0
    Ac[-1] = 0
    o_s, o_c = 2 + 2 * (J + 1) + 3 * int(ps[1]), 2 + 2 * (J + 1) + 3 * int(pc[1])
    deg, cdeg = (int(ps[o_s]), int(pc[o_c])), (int(ps[o_s + 1]), int(pc[o_c + 1]))
    Y = torch.zeros(P, 2 * C).to(dt).cuda()
    nb = ops.semch_agg_blocks(F, C)
    part = torch.zeros(nb, 2 * C, 2).cuda()
    # odd C-products exercise the centred storage (outputs minus a per-channel centre), the others the plain one
    ctr = (rand(gen, C), rand(gen, C)) if (J * C) % 2 == 0 else (None, None)
    ops.semch_agg_fwd(H.cuda(), F, J, C, As.cuda(), dev(ps), Ac.cuda(), dev(pc), Y, part, deg=deg,
                      center=tuple(c.cuda() if c is not None else None for c in ctr))
    Yh = np.zeros((P, 2 * C))
    ph = np.zeros((nb, 2 * C, 2))
    rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
    kc.semch_agg_fwd(host(H), F, J, C, host(As)[:-1], ps, host(Ac)[:-1], pc, Yh, ph, round_fn=rnd,
                     center_sym=host(ctr[0]) if ctr[0] is not None else None, center_con=host(ctr[1]) if ctr[1] is not None else None)
    close(host(Y), Yh, dt, 'agg fwd')
    close(host(part).sum(0), ph.sum(0), dt, 'agg partial totals', fp32=1e-4, bf16=3e-2)
    close(host(part), ph, dt, 'agg partials per block', fp32=1e-4, bf16=3e-2)
    # backward
    dY = rand(gen, P, 2 * C).to(dt)
    dH = torch.full((P, ldh), 5.0).to(dt).cuda()
    ns, nc = int(ps[1]), int(pc[1])
    dA = torch.full((ns + nc, C), 9.0).cuda()
    ws = torch.empty(ops.semch_agg_bwd_ws(F, C, ns, nc)).cuda()
    ops.semch_agg_bwd(dY.cuda(), H.cuda(), F, J, C, As.cuda(), dev(ps), Ac.cuda(), dev(pc), dH, dA, ws, cdeg=cdeg)
    dAs, dAc = dA[:ns], dA[ns:]
    dHh = np.full((P, ldh), 5.0)
    dAsh, dAch = np.zeros((int(ps[1]), C)), np.zeros((int(pc[1]), C))
    kc.semch_agg_bwd(host(dY), host(H), F, J, C, host(As)[:-1], ps, host(Ac)[:-1], pc, dHh, dAsh, dAch, round_fn=rnd)
    got = host(dH)
    close(got[:, :4 * C], dHh[:, :4 * C], dt, 'agg bwd dH')
    assert np.all(got[:, 4 * C:] == 5.0)
    close(host(dAs), dAsh, dt, 'agg bwd dA_sym', fp32=1e-4, bf16=2e-2)
    close(host(dAc), dAch, dt, 'agg bwd dA_con', fp32=1e-4, bf16=2e-2)


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('J,C,F,generic', [(17, 16, 37, False), (19, 128, 50, False), (15, 8, 9, False), (17, 512, 6, False),
                                           (16, 2048, 3, False), (17, 256, 2200, False), (17, 128, 301, True), (19, 256, 7, False),
                                           (15, 128, 1, False)])
def test_attention(ops, J, C, F, generic, dt):
    """head widths 32 / 64 / 128 take the wave-per-unit kernels, everything else (and generic=True) the block-per-head ones"""
    gen = torch.Generator().manual_seed(J + C)
    nh = 4
    P = F * J
    ld = C + 2 * nh + 16
    Hx = rand(gen, P, ld).to(dt)
    Ck = rand(gen, nh, J, J, scale=0.1)
    Hd = Hx.cuda()
    G, AC = Hd[:, :C], Hd[:, C:C + 2 * nh]
    Y = torch.full((P, C + 4), 3.0).to(dt).cuda()
    ops.attn_fwd(G, AC, Ck.cuda(), F, J, C, nh, Y[:, :C])
    Hh = host(Hx)
    Yh = np.zeros((P, C))
    rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
    kc.attn_fwd(Hh[:, :C], Hh[:, C:C + 2 * nh], host(Ck), F, J, C, nh, Yh, round_fn=rnd)
    got = host(Y)
    close(got[:, :C], Yh, dt, 'attn fwd')
    assert np.all(got[:, C:] == 3.0)
    dY = rand(gen, P, C).to(dt)
    dHd = torch.full((P, ld), 2.0).to(dt).cuda()
    dCk = torch.full((nh, J, J), 0.5).cuda()             # accumulated into
    dbias = torch.full((C + 2 * nh,), -0.25).cuda()
    ops.attn_bwd(dY.cuda(), G, AC, Ck.cuda(), F, J, C, nh, dHd[:, :C], dHd[:, C:C + 2 * nh], dCk, dbias=dbias, generic=generic)
    dG, dAC, dCkh, dbh = np.zeros((P, C)), np.zeros((P, 2 * nh)), np.full((nh, J, J), 0.5), np.full(C + 2 * nh, -0.25)
    kc.attn_bwd(host(dY), Hh[:, :C], Hh[:, C:C + 2 * nh], host(Ck), F, J, C, nh, dG, dAC, dCkh, round_fn=rnd, dbias=dbh)
    assert np.abs(host(dbias) - dbh).max() <= (2e-4 if dt == torch.float32 else 2e-2) * max(1.0, np.abs(dbh).max()), 'attn bwd dbias'
    got = host(dHd)
    close(got[:, :C], dG, dt, 'attn bwd dG')
    close(got[:, C:C + 2 * nh], dAC, dt, 'attn bwd dAC', fp32=1e-4, bf16=3e-2)
    assert np.all(got[:, C + 2 * nh:] == 2.0)
    close(host(dCk), dCkh, dt, 'attn bwd dC_k', fp32=1e-4, bf16=3e-2)


@pytest.mark.parametrize('J,C,F', [(17, 128, 300), (19, 256, 41), (15, 512, 7)])
def test_deferred_finishes_equal_the_immediate_ones(ops, J, C, F):
    """gast_attn_bwd_deferred / gast_semch_agg_bwd_deferred + ONE gast_rowsum_multi == gast_attn_bwd / gast_semch_agg_bwd (whose last
    launch is that reduction): dbias, dC_k (accumulated) and dA (overwritten), bit for bit; dH untouched by the deferral."""
    gen = torch.Generator().manual_seed(J * C + F)
    nh, P = 4, F * J
    ps, pc = patterns(J)
    ns, nc = int(ps[1]), int(pc[1])
    o_s, o_c = 2 + 2 * (J + 1) + 3 * ns, 2 + 2 * (J + 1) + 3 * nc
    cdeg = (int(ps[o_s + 1]), int(pc[o_c + 1]))
    ldh = 5 * C + 8
    H = rand(gen, P, ldh).cuda()
    Ck = rand(gen, nh, J, J, scale=0.1).cuda()
    dYa, dY = rand(gen, P, C).cuda(), rand(gen, P, 2 * C).cuda()
    As, Ac = torch.rand(ns + 1, C, generator=gen), torch.rand(nc + 1, C, generator=gen)
    As[-1] = 0
    Ac[-1] = 0
    As, Ac = As.cuda(), Ac.cuda()
    out = {}
    for tag in ('now', 'later'):
        q = [] if tag == 'later' else None
        dH = torch.full((P, ldh), 2.0).cuda()
        dCk, dbias = torch.full((nh, J, J), 0.5).cuda(), torch.full((C + 2 * nh,), -0.25).cuda()
        dA = torch.full((ns + nc, C), 9.0).cuda()
        ws = torch.empty(ops.semch_agg_bwd_ws(F, C, ns, nc)).cuda()
        kw = {} if q is None else {'defer': q}
        ops.attn_bwd(dYa, H[:, 4 * C:5 * C], H[:, 5 * C:], Ck, F, J, C, nh, dH[:, 4 * C:5 * C], dH[:, 5 * C:], dCk, dbias=dbias, **kw)
        ops.semch_agg_bwd(dY, H, F, J, C, As, dev(ps), Ac, dev(pc), dH, dA, ws, cdeg=cdeg, **kw)
        if q is not None:
            assert len(q) == 2, 'both kernels defer their finish at these head widths'
            ops.rowsum_multi(q)
        torch.cuda.synchronize()
        out[tag] = (dH, dCk, dbias, dA)
    for a, b, name in zip(out['now'], out['later'], ('dH', 'dC_k', 'dbias', 'dA')):
        assert torch.equal(a, b), name


# ------------------------------------------------------------------------------------------------ BN / elementwise
def test_bn_finalize_and_backward(ops):
    gen = torch.Generator().manual_seed(3)
    nblk, ncol, col0, N, count = 7, 40, 8, 24, 5000.0
    x = rand(gen, 5000, N) * 2 + 0.7
    part = torch.zeros(nblk, ncol, 2)
    chunks = torch.chunk(x, nblk)
    for b, ch in enumerate(chunks):
        part[b, col0:col0 + N, 0] = ch.sum(0)
        part[b, col0:col0 + N, 1] = (ch * ch).sum(0)
    gamma, beta = torch.rand(N, generator=gen) + 0.5, rand(gen, N)
    rm, rv = rand(gen, N), torch.rand(N, generator=gen) + 0.5
    nbt = torch.tensor(4, dtype=torch.int64)
    outs = [torch.zeros(N).cuda() for _ in range(4)]
    rmd, rvd, nbtd = rm.cuda(), rv.cuda(), nbt.cuda()
    ops.bn_finalize(part.cuda(), nblk, col0, N, count, gamma.cuda(), beta.cuda(), rmd, rvd, nbtd, 0.1, 1e-5, *outs)
    ho = [np.zeros(N) for _ in range(4)]
    rmh, rvh, nbth = host(rm), host(rv), np.array(4)
    kc.bn_finalize(host(part), nblk, col0, N, count, host(gamma), host(beta), rmh, rvh, nbth, 0.1, 1e-5, *ho)
    for a, b, nme in zip(outs, ho, ('scale', 'shift', 'mean', 'rstd')):
        close(host(a), b, torch.float32, 'bn_finalize ' + nme)
    close(host(rmd), rmh, torch.float32, 'running_mean')
    close(host(rvd), rvh, torch.float32, 'running_var')
    assert int(nbtd.item()) == 5
    # torch cross-check of the running stats semantics
    bn = torch.nn.BatchNorm1d(N, momentum=0.1)
    with torch.no_grad():
        bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train(); bn(x)
    close(host(rmd), host(bn.running_mean), torch.float32, 'running_mean vs torch', fp32=1e-5)
    close(host(rvd), host(bn.running_var), torch.float32, 'running_var vs torch', fp32=1e-5)
    # centred storage: statistics of (x - running_mean) must produce the same running stats and an equivalent affine map
    xc = x - rm[None, :]
    partc = torch.zeros(nblk, ncol, 2)
    for b, ch in enumerate(torch.chunk(xc, nblk)):
        partc[b, col0:col0 + N, 0] = ch.sum(0)
        partc[b, col0:col0 + N, 1] = (ch * ch).sum(0)
    outc = [torch.zeros(N).cuda() for _ in range(4)]
    rmc, rvc, nbtc = rm.cuda(), rv.cuda(), nbt.cuda()
    ops.bn_finalize(partc.cuda(), nblk, col0, N, count, gamma.cuda(), beta.cuda(), rmc, rvc, nbtc, 0.1, 1e-5, *outc, centered=True)
    hc = [np.zeros(N) for _ in range(4)]
    rmh2, rvh2, nbth2 = host(rm), host(rv), np.array(4)
    kc.bn_finalize(host(partc), nblk, col0, N, count, host(gamma), host(beta), rmh2, rvh2, nbth2, 0.1, 1e-5, *hc, centered=True)
    for a, b, nme in zip(outc, hc, ('scale', 'shift', 'mean', 'rstd')):
        close(host(a), b, torch.float32, 'centred bn_finalize ' + nme)
    close(host(rmc), host(bn.running_mean), torch.float32, 'centred running_mean vs torch', fp32=1e-5)
    close(host(rvc), host(bn.running_var), torch.float32, 'centred running_var vs torch', fp32=1e-5)
    # scale * (x - rm) + shift_c == scale * x + shift
    close(host(outc[0]), host(outs[0]), torch.float32, 'centred scale', fp32=1e-5)
    close(host(outc[1]) - host(outc[0]) * host(rm), host(outs[1]), torch.float32, 'centred shift', fp32=1e-4)
    # eval
    sc, sh = torch.zeros(N).cuda(), torch.zeros(N).cuda()
    ops.bn_eval(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5, N, sc, sh, centered=True)
    close(host(sh), host(beta), torch.float32, 'centred bn_eval shift')
    ops.bn_eval(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5, N, sc, sh)
    sch, shh = np.zeros(N), np.zeros(N)
    kc.bn_eval(host(gamma), host(beta), host(rm), host(rv), 1e-5, N, sch, shh)
    close(host(sc), sch, torch.float32, 'bn_eval scale'); close(host(sh), shh, torch.float32, 'bn_eval shift')
    # backward finalize
    mean, rstd = outs[2], outs[3]
    bo = [torch.zeros(N).cuda() for _ in range(5)]
    ops.bn_bwd_finalize(part.cuda(), nblk, col0, N, count, gamma.cuda(), mean, rstd, *bo)
    bh = [np.zeros(N) for _ in range(5)]
    kc.bn_bwd_finalize(host(part), nblk, col0, N, count, host(gamma), host(mean), host(rstd), *bh)
    for a, b, nme in zip(bo, bh, ('dgamma', 'dbeta', 'ka', 'kb', 'kc')):
        close(host(a), b, torch.float32, 'bn_bwd_finalize ' + nme, fp32=1e-4)


def test_bn_finalize_multi(ops):
    """three BatchNorms of different widths (two of them slices of one partial buffer, like bn_1 | bn_2) in one launch, forward
    and backward finalizes, against the single-job contract"""
    gen = torch.Generator().manual_seed(11)
    nblk = 9
    def partial_sums(rows, ncol):
        x = rand(gen, rows, ncol) * 2 + 0.5
        pt = torch.zeros(nblk, ncol, 2)
        for b, ch in enumerate(torch.chunk(x, nblk)):
            pt[b, :, 0] = ch.sum(0)
            pt[b, :, 1] = (ch * ch).sum(0)
        return pt
    pA, pB = partial_sums(777, 96), partial_sums(123, 40)
    specs = [(pA, 0, 64, 777.0), (pA, 64, 32, 777.0), (pB, 4, 36, 123.0)]
    dev_jobs, host_jobs, bwd_dev, bwd_host = [], [], [], []
    for pt, col0, N, count in specs:
        gamma, beta = torch.rand(N, generator=gen) + 0.5, rand(gen, N)
        rm, rv = rand(gen, N), torch.rand(N, generator=gen) + 0.5
        d = dict(partials=pt.cuda(), nblk=nblk, col0=col0, N=N, count=count, gamma=gamma.cuda(), beta=beta.cuda(), running_mean=rm.cuda(),
                 running_var=rv.cuda(), nbt=torch.tensor(2, dtype=torch.int64).cuda(), momentum=0.1, eps=1e-5,
                 scale=torch.zeros(N).cuda(), shift=torch.zeros(N).cuda(), mean=torch.zeros(N).cuda(), rstd=torch.zeros(N).cuda())
        h = dict(partials=host(pt), nblk=nblk, col0=col0, N=N, count=count, gamma=host(gamma), beta=host(beta), running_mean=host(rm),
                 running_var=host(rv), nbt=np.array(2), momentum=0.1, eps=1e-5, scale=np.zeros(N), shift=np.zeros(N), mean=np.zeros(N),
                 rstd=np.zeros(N))
        dev_jobs.append(d)
        host_jobs.append(h)
    ops.bn_finalize_multi(dev_jobs)
    for d, h in zip(dev_jobs, host_jobs):
        kc.bn_finalize(**h)
        for k in ('scale', 'shift', 'mean', 'rstd', 'running_mean', 'running_var'):
            close(host(d[k]), h[k], torch.float32, 'bn_finalize_multi ' + k, fp32=1e-5)
        assert int(d['nbt'].item()) == 3
        N = d['N']
        bd = dict(partials=d['partials'], nblk=nblk, col0=d['col0'], N=N, count=d['count'], gamma=d['gamma'], mean=d['mean'], rstd=d['rstd'],
                  dgamma=torch.zeros(N).cuda(), dbeta=torch.zeros(N).cuda(), ka=torch.zeros(N).cuda(), kb=torch.zeros(N).cuda(),
                  kc=torch.zeros(N).cuda())
        bh = dict(partials=h['partials'], nblk=nblk, col0=h['col0'], N=N, count=h['count'], gamma=h['gamma'], mean=h['mean'], rstd=h['rstd'],
                  dgamma=np.zeros(N), dbeta=np.zeros(N), ka=np.zeros(N), kb=np.zeros(N), kc=np.zeros(N))
        bwd_dev.append(bd)
        bwd_host.append(bh)
    ops.bn_bwd_finalize_multi(bwd_dev)
    for bd, bh in zip(bwd_dev, bwd_host):
        kc.bn_bwd_finalize(**bh)
        for k in ('dgamma', 'dbeta', 'ka', 'kb'):
            close(host(bd[k]), bh[k], torch.float32, 'bn_bwd_finalize_multi ' + k, fp32=1e-4)
        # kc = -kb*mean - ka*dbeta/count cancels almost completely on this data: compare against the size of its terms
        terms = np.abs(bh['kb'] * host(bd['mean'])) + np.abs(bh['ka'] * bh['dbeta'] / bd['count'])
        assert np.all(np.abs(host(bd['kc']) - bh['kc']) <= 1e-5 * terms + 1e-12)
    # fused finalize + apply (short tensors): same jobs, dgamma / dbeta accumulated onto a fill, dz rewritten in place
    for dt in DTYPES:
        rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
        fj, refs = [], []
        for bd, bh in zip(bwd_dev, bwd_host):
            N, rows = bd['N'], 301
            dz, X = rand(gen, rows, N + 8).to(dt), rand(gen, rows, N + 4).to(dt)
            dzd = dz.cuda()
            j = dict(bd)
            j.update(dgamma=torch.full((N,), 0.5).cuda(), dbeta=torch.full((N,), -1.0).cuda(), accumulate=True, dz=dzd[:, :N],
                     X=X.cuda()[:, :N], rows=rows)
            for k in ('ka', 'kb', 'kc'):
                j.pop(k)
            fj.append(j)
            dzh = host(dz).copy()
            kc.bn_bwd_apply(dzh, host(X), rows, N, bh['ka'], bh['kb'], bh['kc'], round_fn=rnd)
            refs.append((dzd, dzh, bh))
        ops.bn_bwd_fused_multi(fj)
        for j, (dzd, dzh, bh) in zip(fj, refs):
            N = j['N']
            close(host(dzd)[:, :N], dzh[:, :N], dt, 'bn_bwd_fused dz')
            assert np.array_equal(host(dzd)[:, N:], dzh[:, N:]), 'wrote outside the N columns'
            close(host(j['dgamma']) - 0.5, bh['dgamma'], torch.float32, 'bn_bwd_fused dgamma', fp32=1e-4)
            close(host(j['dbeta']) + 1.0, bh['dbeta'], torch.float32, 'bn_bwd_fused dbeta', fp32=1e-4)


@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('B,T,J,N,frames', [(7, 19, 17, 512, (0, 9, 18)), (3, 64, 15, 64, (63,)), (2, 5, 19, 8, (0, 1, 2, 3, 4)), (4, 3, 17, 1032, ())])
def test_bn_bwd_apply_frames(ops, B, T, J, N, frames, dt):
    """gast_bn_bwd_apply_frames: the rows of the frames outside the mask are NaN here -- they must not be read -- and come out as kb*x + kc;
    a full mask equals gast_bn_bwd_apply, an empty one is a pure function of X."""
    gen = torch.Generator().manual_seed(B * 100 + T)
    rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
    rows, ld = B * T * J, N + 8
    mask = sum(1 << t for t in frames)
    X = rand(gen, rows, ld).to(dt)
    ka, kb, kcc = rand(gen, N), rand(gen, N, scale=0.1), rand(gen, N, scale=0.1)
    dz = rand(gen, rows, ld).to(dt)
    t_of = (torch.arange(rows) // J) % T
    dead = torch.tensor([t not in frames for t in range(T)])[t_of]
    dzd = dz.clone()
    dzd[dead] = float('nan')
    dzd = dzd.cuda()
    ops.bn_bwd_apply_frames(dzd[:, :N], X.cuda()[:, :N], rows, N, ka.cuda(), kb.cuda(), kcc.cuda(), T, J, mask)
    dzh = host(dz)
    kc.bn_bwd_apply_frames(dzh[:, :N], host(X)[:, :N], rows, N, host(ka), host(kb), host(kcc), T, J, mask, round_fn=rnd)
    got = host(dzd)
    assert np.isfinite(got[:, :N]).all(), 'a dead frame was read'
    close(got[:, :N], dzh[:, :N], dt, 'bn_bwd_apply_frames')
    assert np.isnan(got[dead.numpy()][:, N:]).all() or not dead.any()          # (the columns past N are untouched)


@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('rows,K', [(2176, 1024), (2175, 512), (1, 64), (33, 2048 + 8), (70, 8)])
def test_shrink_rowwise(ops, rows, K, dt):
    """gast_shrink_fwd / gast_shrink_bwd (reference gast_net.py:99,176-178) against the numpy contract: full size of BASELINE configs[1]
    (B*J = 2176 rows, 1024 channels), an odd row count (the wave that owns one row, the ragged last row block), a single row, a K that
    is not a multiple of the 256-column block, K = 8."""
    gen = torch.Generator().manual_seed(rows * 7 + K)
    rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
    ld = K + 8
    O = rand(gen, rows, ld).to(dt)
    W = rand(gen, 3, K, scale=0.2).to(dt)
    sc, sh = torch.rand(K, generator=gen) + 0.5, rand(gen, K, scale=0.5)
    pred = torch.full((rows, 3), 7.0).cuda()
    Od = O.cuda()
    ops.shrink_fwd(Od[:, :K], rows, K, sc.cuda(), sh.cuda(), W.cuda(), pred)
    ph = np.zeros((rows, 3))
    kc.shrink_fwd(host(O), rows, K, host(sc), host(sh), host(W), ph)
    close(host(pred), ph, torch.float32, 'shrink_fwd', fp32=2e-5)          # (fp32 output and accumulation in both storage types)
    dp = torch.zeros(rows, 8)
    dp[:, :3] = rand(gen, rows, 3)
    dp = dp.to(dt)
    dO = torch.full((rows, ld), 4.0).to(dt).cuda()
    nb = ops.shrink_bwd_blocks(rows)
    assert nb == kc.shrink_bwd_blocks(rows)
    part = torch.full((nb, K, 2), 5.0).cuda()            # fully overwritten
    ops.shrink_bwd(dp.cuda(), W.cuda(), Od[:, :K], rows, K, sc.cuda(), sh.cuda(), dO[:, :K], part)
    dh = np.full((rows, ld), 4.0)
    pth = np.zeros((nb, K, 2))
    kc.shrink_bwd(host(dp), host(W), host(O), rows, K, host(sc), host(sh), dh, pth, round_fn=rnd)
    close(host(dO), dh, dt, 'shrink_bwd dO')
    close(host(part), pth, dt, 'shrink_bwd partials', fp32=1e-4, bf16=3e-2)


def test_shrink_rejects_bad_arguments(ops):
    O = torch.zeros(8, 64).cuda()
    W = torch.zeros(3, 64).cuda()
    v = torch.ones(64).cuda()
    pred = torch.zeros(8, 3).cuda()
    with pytest.raises(RuntimeError):
        ops.shrink_fwd(O[:, :62], 8, 62, v, v, W, pred)                      # K % 4
    with pytest.raises(RuntimeError):
        ops.shrink_fwd(O, 8, 64, v, v, torch.zeros(5, 64).cuda(), torch.zeros(8, 5).cuda())      # D > 4


@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('rows,N', [(301, 16), (1000, 128), (77, 2048 + 64), (5000, 8)])
def test_rowwise_kernels(ops, rows, N, dt):
    from gast_hip.binding import Dropout, dropout_params
    gen = torch.Generator().manual_seed(rows + N)
    rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
    ld = N + 8
    X = rand(gen, rows, ld).to(dt)
    sc, sh = torch.rand(N, generator=gen) + 0.5, rand(gen, N, scale=0.5)
    # bnrelu_apply
    Y = torch.full((rows, ld), 9.0).to(dt).cuda()
    ops.bnrelu_apply(X.cuda()[:, :N], rows, N, sc.cuda(), sh.cuda(), Y[:, :N])
    Yh = np.full((rows, ld), 9.0)
    kc.bnrelu_apply(host(X)[:, :N], rows, N, host(sc), host(sh), Yh[:, :N], round_fn=rnd)
    close(host(Y), Yh, dt, 'bnrelu_apply')
    # bn_bwd_apply
    ka, kb, kcc = rand(gen, N), rand(gen, N, scale=0.1), rand(gen, N, scale=0.1)
    dz = rand(gen, rows, ld).to(dt)
    dzd = dz.cuda().clone()
    ops.bn_bwd_apply(dzd[:, :N], X.cuda()[:, :N], rows, N, ka.cuda(), kb.cuda(), kcc.cuda())
    dzh = host(dz)
    kc.bn_bwd_apply(dzh[:, :N], host(X)[:, :N], rows, N, host(ka), host(kb), host(kcc), round_fn=rnd)
    close(host(dzd), dzh, dt, 'bn_bwd_apply')
    # bnrelu_bwd_mask (+dropout)
    thresh, inv_keep = dropout_params(0.2)
    dY = rand(gen, rows, ld).to(dt)
    out = torch.full((rows, ld), 4.0).to(dt).cuda()
    nb = ops.rowwise_blocks(rows, N)
    part = torch.zeros(nb, N, 2).cuda()
    Xd = X.cuda()
    ops.bnrelu_bwd_mask(dY.cuda()[:, :N], Xd[:, :N], rows, N, sc.cuda(), sh.cuda(), True, 5, Dropout(seed_tensor(42), thresh, inv_keep),
                        out[:, :N], part)
    outh = np.full((rows, ld), 4.0)
    ph = np.zeros((nb, N, 2))
    hX = host(X)
    kc.bnrelu_bwd_mask(host(dY)[:, :N], hX[:, :N], rows, N, host(sc), host(sh), True, 5, (42, thresh, inv_keep), outh[:, :N], ph, round_fn=rnd)
    close(host(out), outh, dt, 'bnrelu_bwd_mask dz')
    close(host(part), ph, dt, 'bnrelu_bwd_mask partials', fp32=1e-4, bf16=3e-2)
    # colsum
    cs = torch.full((N,), 1.0).cuda()
    ops.colsum(Xd[:, :N], rows, N, cs, zero_first=False)
    close(host(cs), 1.0 + hX[:, :N].sum(0), dt, 'colsum', fp32=1e-4, bf16=1e-2)


@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
def test_residual_fwd(ops, dt):
    from gast_hip.binding import Dropout, dropout_params
    gen = torch.Generator().manual_seed(8)
    rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
    B, Tp, Tn, J, N = 3, 11, 5, 17, 64
    O = rand(gen, B * Tp * J, N).to(dt)
    T2 = rand(gen, B * Tn * J, N).to(dt)
    scO, shO = torch.rand(N, generator=gen) + 0.5, rand(gen, N, scale=0.3)
    sc2, sh2 = torch.rand(N, generator=gen) + 0.5, rand(gen, N, scale=0.3)
    thresh, inv_keep = dropout_params(0.3)
    for omap in (kc.RowMap(Tp, 1, 3), kc.RowMap(Tp, 2, 1)):
        Xn = torch.zeros(B * Tn * J, N).to(dt).cuda()
        ops.residual_fwd(O.cuda(), omap, scO.cuda(), shO.cuda(), T2.cuda(), sc2.cuda(), sh2.cuda(), True, 6,
                         Dropout(seed_tensor(11), thresh, inv_keep), B, Tn, J, N, Xn)
        Xh = np.zeros((B * Tn * J, N))
        kc.residual_fwd(host(O), omap, host(scO), host(shO), host(T2), host(sc2), host(sh2), True, 6, (11, thresh, inv_keep), B, Tn, J, N, Xh,
                        round_fn=rnd)
        close(host(Xn), Xh, dt, 'residual_fwd')


# ------------------------------------------------------------------------------------------------ input side
@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('B,T,J,k0,ts,C', [(3, 29, 17, 3, 1, 16), (5, 27, 17, 3, 3, 128), (2, 17, 19, 5, 1, 32), (2, 15, 15, 5, 5, 8)])
def test_input_side(ops, B, T, J, k0, ts, C, dt):
    gen = torch.Generator().manual_seed(B * T)
    rnd = (lambda v: v) if dt == torch.float32 else (lambda v: host(torch.from_numpy(v).to(H16)))
    F_in = 2
    x = torch.rand(B, T, J, F_in, generator=gen) * 2 - 1
    rows = B * T * J
    nb = ops.input_stats_blocks(rows)
    part = torch.zeros(nb, F_in, 2).cuda()
    ops.input_stats(x.cuda(), rows, F_in, part)
    ph = np.zeros((nb, F_in, 2))
    kc.input_stats(host(x), rows, F_in, ph)
    close(host(part), ph, torch.float32, 'input_stats', fp32=1e-5)
    W = rand(gen, C, F_in, k0, 1)
    sc0, sh0 = torch.rand(F_in, generator=gen) + 0.5, rand(gen, F_in, scale=0.2)
    T_out = (T - k0) // ts + 1
    P = B * T_out * J
    E = torch.zeros(P, C).to(dt).cuda()
    nbe = ops.rowwise_blocks(P, C)
    pe = torch.zeros(nbe, C, 2).cuda()
    ctr = rand(gen, C) if k0 == 3 else None
    ops.expand_fwd(x.cuda(), B, T, J, F_in, k0, ts, W.cuda(), sc0.cuda(), sh0.cuda(), C, E, pe, center=ctr.cuda() if ctr is not None else None)
    Eh = np.zeros((P, C))
    peh = np.zeros((nbe, C, 2))
    kc.expand_fwd(host(x), B, T, J, F_in, k0, ts, host(W), host(sc0), host(sh0), C, Eh, peh, round_fn=rnd,
                  center=host(ctr) if ctr is not None else None)
    close(host(E), Eh, dt, 'expand_fwd')
    close(host(pe), peh, dt, 'expand_fwd partials', fp32=1e-4, bf16=3e-2)
    dE = rand(gen, P, C).to(dt)
    mean0, rstd0 = rand(gen, F_in, scale=0.1), torch.rand(F_in, generator=gen) + 1.0
    g0, b0 = torch.rand(F_in, generator=gen) + 0.5, rand(gen, F_in, scale=0.3)
    dW = torch.full((C, F_in, k0, 1), 3.0).cuda()
    dg0, db0 = torch.full((F_in,), 0.25).cuda(), torch.full((F_in,), -0.5).cuda()     # accumulated into
    ops.expand_bwd(dE.cuda(), x.cuda(), B, T, J, F_in, k0, ts, mean0.cuda(), rstd0.cuda(), C, W.cuda(), g0.cuda(), b0.cuda(), dW, dg0, db0)
    dWh, dgh, dbh = np.zeros((C, F_in, k0, 1)), np.full(F_in, 0.25), np.full(F_in, -0.5)
    kc.expand_bwd(host(dE), host(x), B, T, J, F_in, k0, ts, host(mean0), host(rstd0), C, host(W), host(g0), host(b0), dWh, dgh, dbh)
    close(host(dW), dWh, dt, 'expand_bwd dW', fp32=1e-4, bf16=1e-2)
    close(host(dg0), dgh, dt, 'expand_bwd dgamma0', fp32=1e-4, bf16=1e-2)
    close(host(db0), dbh, dt, 'expand_bwd dbeta0', fp32=1e-4, bf16=1e-2)


# ------------------------------------------------------------------------------------------------ pack / unpack
@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
def test_pack_unpack_tables(ops, dt):
    """gast_strided_copy / gast_fold / gast_unfold driven by the real job tables of a model, against the tensor-view mirror."""
    import copy
    from fake_backend import OracleOps
    from gast_hip.packer import Packer
    from model.gast_net import SpatioTemporalModel
    from oracle.gast_oracle import adj_from_parents
    torch.manual_seed(3)
    m = SpatioTemporalModel(torch.from_numpy(adj_from_parents(PARENTS[17])), 17, 2, 17, filter_widths=[3, 3, 3], channels=32)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    mg = copy.deepcopy(m).cuda()
    pk_c, pk_g = Packer(m, m._runner.spec), Packer(mg, mg._runner.spec)
    st_c, st_g = pk_c.state(torch.device('cpu'), torch.float32), pk_g.state(torch.device('cuda'), dt)
    mirror = OracleOps()
    mirror.run_pack(pk_c, st_c)
    ops.run_pack(pk_g, st_g)
    torch.cuda.synchronize()
    close(host(st_g['Wb']), host(st_c['Wb']), dt, 'packed operands', fp32=1e-6, bf16=8e-3)
    close(host(st_g['Fb']), host(st_c['Fb']), torch.float32, 'packed fp32 values', fp32=1e-6)
    gen = torch.Generator().manual_seed(4)
    Sb = torch.randn(pk_c.S.size, generator=gen)
    for acc in (False, True):
        Gc = torch.full((pk_c.gsize,), 0.5)
        Gg = torch.full((pk_g.gsize,), 0.5).cuda()
        mirror.run_unpack(pk_c, st_c, Sb, Gc, acc)
        ops.run_unpack(pk_g, st_g, Sb.cuda(), Gg, acc)
        torch.cuda.synchronize()
        close(host(Gg), host(Gc), torch.float32, 'unpacked gradients acc=%s' % acc, fp32=2e-5)


# ------------------------------------------------------------------------------------------------ streaming frame windows
def test_stream_shift_multi(ops):
    """every frame window of the causal stream advances by one frame in ONE launch (in place), vs the contract"""
    gen = torch.Generator().manual_seed(31)
    jobs, hosts = [], []
    for (B, Tb, X, ldnew) in ((2, 3, 17 * 32, 17 * 32), (4, 7, 17 * 64, 17 * 64 + 8), (1, 19, 17 * 128, 17 * 128), (2, 1, 68, 68)):
        buf = rand(gen, B, Tb, X)
        new = rand(gen, B, ldnew)
        jobs.append((buf.cuda(), new.cuda()[:, :X]))
        hb = buf.numpy().copy()
        kc.stream_shift(hb, new.numpy()[:, :X])
        hosts.append(hb)
    ops.stream_shift_multi(jobs)
    torch.cuda.synchronize()
    for (b, _), hb in zip(jobs, hosts):
        assert np.array_equal(b.cpu().numpy(), hb)


# ------------------------------------------------------------------------------------------------ BatchNorm backward fused into its one reader (round 5)
def _bn_jobs(gen, specs, nblk=37):
    """forward finalize jobs over random partial sums; specs = [(col0, N)] slices of ONE partial buffer (like bn_1 | bn_2)"""
    ncol = max(c0 + n for c0, n in specs) + 4
    rows = 4321
    x = rand(gen, rows, ncol) * 1.5 + 0.3
    pt = torch.zeros(nblk, ncol, 2)
    for b, ch in enumerate(torch.chunk(x, nblk)):
        pt[b, :, 0] = ch.sum(0)
        pt[b, :, 1] = (ch * ch).sum(0)
    ptd = pt.cuda()
    tot = sum(n for _, n in specs)
    jobs = []
    for c0, n in specs:
        jobs.append(dict(partials=ptd, nblk=nblk, col0=c0, N=n, count=float(rows), gamma=(torch.rand(n, generator=gen) + 0.5).cuda(),
                         beta=rand(gen, n).cuda(), running_mean=rand(gen, n).cuda(), running_var=(torch.rand(n, generator=gen) + 0.5).cuda(),
                         nbt=torch.tensor(5, dtype=torch.int64).cuda(), momentum=0.1, eps=1e-5))
    return jobs, tot


@pytest.mark.parametrize('J,C,F', [(17, 256, 700), (19, 128, 50), (17, 64, 3), (15, 32, 41)])
def test_agg_bwd_with_the_batchnorm_backward_fused(ops, J, C, F):
    """gast_semch_agg_bwd_bn: the aggregation backward applies ka*dY + kb*Y + kc while it stages dY: dH, dA and the BatchNorm parameter gradients are bit-equal to finalize launch + apply launch + aggregation launch"""
    gen = torch.Generator().manual_seed(J * C + F)
    ps, pc = patterns(J)
    P = F * J
    ldh = 5 * C + 8
    H = rand(gen, P, ldh).cuda()
    As, Ac = torch.rand(int(ps[1]) + 1, C, generator=gen), torch.rand(int(pc[1]) + 1, C, generator=gen)
    As[-1] = 0
    Ac[-1] = 0
    As, Ac = As.cuda(), Ac.cuda()
    o_s, o_c = 2 + 2 * (J + 1) + 3 * int(ps[1]), 2 + 2 * (J + 1) + 3 * int(pc[1])
    cdeg = (int(ps[o_s + 1]), int(pc[o_c + 1]))
    if not ops.semch_agg_bwd_fuses_bn(H, F, J, C, As, Ac, cdeg):
        pytest.skip('this shape takes the kernel without LDS staging')
    fj, N = _bn_jobs(gen, [(0, C), (C, C)])
    dY0, Yp = rand(gen, P, 2 * C).cuda(), rand(gen, P, 2 * C).cuda()
    mean, rstd = rand(gen, N).cuda(), (torch.rand(N, generator=gen) + 0.5).cuda()
    ns, nc = int(ps[1]), int(pc[1])
    outs = []
    for fused in (False, True):
        o, jobs = 0, []
        kabc = torch.full((3, N), float('nan')).cuda()
        for j in fj:
            n = j['N']
            jobs.append(dict(partials=j['partials'], nblk=j['nblk'], col0=j['col0'], N=n, count=j['count'], gamma=j['gamma'], mean=mean[o:o + n],
                             rstd=rstd[o:o + n], dgamma=torch.full((n,), 0.25).cuda(), dbeta=torch.full((n,), -0.5).cuda(),
                             ka=kabc[0, o:o + n], kb=kabc[1, o:o + n], kc=kabc[2, o:o + n], accumulate=True))
            o += n
        dY = dY0.clone()
        dH = torch.full((P, ldh), 5.0).cuda()
        dA = torch.full((ns + nc, C), 9.0).cuda()
        ws = torch.empty(ops.semch_agg_bwd_ws(F, C, ns, nc)).cuda()
        if fused:
            ops.bn_bwd_finalize_multi(jobs)
            ops.semch_agg_bwd(dY, H, F, J, C, As, dev(ps), Ac, dev(pc), dH, dA, ws, cdeg=cdeg, bn=(Yp, kabc[0], kabc[1], kabc[2]))
            assert torch.equal(dY, dY0), 'the fused form must not rewrite dY'
        else:
            ops.bn_bwd_finalize_multi(jobs)
            ops.bn_bwd_apply(dY, Yp, P, N, kabc[0], kabc[1], kabc[2])
            ops.semch_agg_bwd(dY, H, F, J, C, As, dev(ps), Ac, dev(pc), dH, dA, ws, cdeg=cdeg)
        torch.cuda.synchronize()
        outs.append((jobs, kabc, dH, dA))
    _assert_bit_equal(outs[1][1], outs[0][1], 'ka / kb / kc')
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a['dgamma'], b['dgamma']) and torch.equal(a['dbeta'], b['dbeta'])
    _assert_bit_equal(outs[1][2], outs[0][2], 'dH of the aggregation backward with the BatchNorm backward fused')
    _assert_bit_equal(outs[1][3], outs[0][3], 'dA of the aggregation backward with the BatchNorm backward fused')


@pytest.mark.parametrize('dt', DTYPES, ids=['f32', 'bf16'])
def test_expand_bwd_with_the_batchnorm_backward_fused(ops, dt):
    """gast_expand_bwd_bn: dz = ka*dE + kb*E + kc applied while dE is loaded -- parameter
    gradients bit-equal to finalize launch + apply launch + expand_bwd launch"""
    gen = torch.Generator().manual_seed(31)
    B, T_in, J, F_in, k0, C = 7, 9, 17, 2, 3, 64
    T_out = T_in - k0 + 1
    P0 = B * T_out * J
    x = rand(gen, B, T_in, J, F_in).cuda()
    dE0, E = rand(gen, P0, C).to(dt).cuda(), rand(gen, P0, C).to(dt).cuda()
    fj, N = _bn_jobs(gen, [(0, C)])
    mean, rstd = rand(gen, C).cuda(), (torch.rand(C, generator=gen) + 0.5).cuda()
    mean0, rstd0 = rand(gen, F_in).cuda(), (torch.rand(F_in, generator=gen) + 0.5).cuda()
    W, g0, b0 = rand(gen, C, F_in, k0).cuda(), rand(gen, F_in).cuda(), rand(gen, F_in).cuda()
    outs = []
    for fused in (False, True):
        kabc = torch.full((3, C), float('nan')).cuda()
        j = fj[0]
        job = dict(partials=j['partials'], nblk=j['nblk'], col0=0, N=C, count=j['count'], gamma=j['gamma'], mean=mean, rstd=rstd,
                   dgamma=torch.full((C,), 0.25).cuda(), dbeta=torch.full((C,), -0.5).cuda(), ka=kabc[0], kb=kabc[1], kc=kabc[2], accumulate=True)
        dE = dE0.clone()
        dW, dg0, db0 = torch.zeros(C, F_in, k0).cuda(), torch.zeros(F_in).cuda(), torch.zeros(F_in).cuda()
        if fused:
            ops.bn_bwd_finalize_multi([job])
            ops.expand_bwd(dE, x, B, T_in, J, F_in, k0, 1, mean0, rstd0, C, W, g0, b0, dW, dg0, db0, bn=(E, kabc[0], kabc[1], kabc[2]))
        else:
            ops.bn_bwd_finalize_multi([job])
            ops.bn_bwd_apply(dE, E, P0, C, kabc[0], kabc[1], kabc[2])
            ops.expand_bwd(dE, x, B, T_in, J, F_in, k0, 1, mean0, rstd0, C, W, g0, b0, dW, dg0, db0)
        torch.cuda.synchronize()
        outs.append((job, kabc, dW, dg0, db0))
    _assert_bit_equal(outs[1][1], outs[0][1], 'ka / kb / kc')
    assert torch.equal(outs[0][0]['dgamma'], outs[1][0]['dgamma']) and torch.equal(outs[0][0]['dbeta'], outs[1][0]['dbeta'])
    _assert_bit_equal(outs[1][2].view(C, -1), outs[0][2].view(C, -1), 'dW of the expand conv with the BatchNorm backward fused')
    # (dgamma0 / dbeta0: the finish pass adds 16 x 3 block sums per address with atomics)
    close(host(outs[1][3]), host(outs[0][3]), torch.float32, 'dgamma0', fp32=1e-5)
    close(host(outs[1][4]), host(outs[0][4]), torch.float32, 'dbeta0', fp32=1e-5)
