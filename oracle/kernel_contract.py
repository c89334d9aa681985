"""Kernel-level CPU restatement (numpy) of what each `gast_*` C-ABI entry point computes -- TEST INFRASTRUCTURE ONLY.

`include/gast_hip.h` is the contract; this file restates it op by op in float64 numpy so that
(a) tests/test_kernels_gpu.py can check every HIP kernel against it on seeded inputs, and
(b) tests/fake_backend.py can run the product's host-side plan on CPU tensors, which pins the *composition* of these
    ops against the reference-generated golden fixtures (tests/test_plan_cpu.py).
The formulas are the ones of SURVEY.md App. A, i.e. algebraic restatements of reference model/local_attention.py:35-53,
model/global_attention.py:52-82 and nn.BatchNorm2d / ReLU / Dropout / Conv2d as used in model/gast_net.py.

Nothing in the product imports this module.
"""
from collections import namedtuple

import numpy as np

RowMap = namedtuple('RowMap', 'T_total t_stride t_off')

PRO_NONE, PRO_BNRELU, PRO_BNRELU_DROP = 0, 1, 2
EPI_PLAIN, EPI_STATS, EPI_BNRELU_BWD = 0, 1, 2
GEMM_BM = 128


# ------------------------------------------------------------------------------------------------ dropout stream
def hash32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xffffffff
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x


def drop_key(seed, salt):
    return (int(seed) * 0x9E3779B9 + int(salt) * 0x85EBCA6B) & 0xffffffff


def drop_mul(key, thresh, inv_keep, e):
    """multiplier (0 or inv_keep) for linear element offsets e (array)."""
    e = np.asarray(e, dtype=np.uint64)
    h = hash32((e >> 1) ^ np.uint64(key))
    bits = np.where((e & 1) == 1, h >> 16, h & 0xffff)
    return np.where(bits >= thresh, np.float64(inv_keep), 0.0)


def dropout_params(p):
    thresh = int(round(p * 65536))
    inv_keep = 65536.0 / (65536 - thresh) if thresh else 1.0
    return thresh, inv_keep


# ------------------------------------------------------------------------------------------------ row maps
def map_rows(rm, B, Tn, J):
    """rows (int64, -1 = invalid) addressed by every m of the (B,Tn,J) domain."""
    b, t, j = np.meshgrid(np.arange(B), np.arange(Tn), np.arange(J), indexing='ij')
    ts = t * rm.t_stride + rm.t_off
    rows = (b * rm.T_total + ts) * J + j
    rows = np.where((ts >= 0) & (ts < rm.T_total), rows, -1)
    return rows.reshape(-1)


def _gather(A, rows, K):
    out = np.zeros((rows.shape[0], K), dtype=np.float64)
    ok = rows >= 0
    out[ok] = A[rows[ok], :K]
    return out, ok


def _prologue(vals, ok, rows, lda, K, pro, scale, shift, salt, drop):
    if pro == PRO_NONE:
        return vals
    y = np.maximum(vals * np.asarray(scale, np.float64)[:K] + np.asarray(shift, np.float64)[:K], 0.0)
    if pro == PRO_BNRELU_DROP and drop is not None and drop[1] != 0:
        seed, thresh, inv_keep = drop
        e = rows[:, None].astype(np.int64) * lda + np.arange(K)[None, :]
        mul = drop_mul(drop_key(seed, salt), thresh, inv_keep, np.where(ok[:, None], e, 0))
        y = y * mul
    y[~ok] = 0.0
    return y


def _ld(a):
    return a.strides[0] // a.itemsize


# ------------------------------------------------------------------------------------------------ gemm / wgrad
def quant_e4m3(x):
    """OCP FP8 E4M3 (exponent bias 7, largest value 448, subnormal step 2^-9; the gfx950 format of v_cvt_pk_fp8_f32 /
    v_mfma_f32_32x32x16_fp8_fp8): round to nearest even, saturating."""
    x = np.asarray(x, np.float64)
    a = np.minimum(np.abs(x), 448.0)
    e = np.clip(np.floor(np.log2(np.maximum(a, 2.0 ** -30))), -6, 8)
    q = 2.0 ** (e - 3)
    return np.sign(x) * np.round(a / q) * q


def f8_weight_scale(W):
    """gast_f8_scale_multi: 2^floor(log2(448 / max|W|)) (1 for an all-zero tensor)"""
    m = float(np.abs(np.asarray(W, np.float64)).max())
    return 1.0 if m == 0.0 else float(2.0 ** np.clip(np.floor(np.log2(448.0 / m)), -40, 40))


def gemm(dom, N, segs, C, cmap, bias=None, addend=None, addmap=None, epi=EPI_PLAIN, partials=None, X=None,
         xscale=None, xshift=None, xdrop=False, xsalt=0, drop=None, round_fn=None, bias_neg=False, f8_scale=None):
    """segs: list of dicts {A, K, map, W, pro, scale, shift, salt}.  drop = (seed, thresh, inv_keep) or None.
    Writes C (and partials) in place.  round_fn emulates storage rounding (identity for fp32).  f8_scale = s: the fp8-operand mode
    of gast_gemm_args.f8_scale -- activations (after the prologue, rounded to their bf16 storage type) and s * weights are rounded to
    e4m3, the products accumulate exactly and the sum is multiplied by 1 / s."""
    B, Tn, J = dom
    M = B * Tn * J
    acc = np.zeros((M, N), dtype=np.float64)
    for s in segs:
        rows = map_rows(s['map'], B, Tn, J)
        a, ok = _gather(s['A'], rows, s['K'])
        a = _prologue(a, ok, rows, _ld(s['A']), s['K'], s.get('pro', 0), s.get('scale'), s.get('shift'), s.get('salt', 0), drop)
        w = np.asarray(s['W'][:N, :s['K']], np.float64)
        if f8_scale is not None:
            if round_fn is not None and s.get('pro', 0):
                a = round_fn(a)
            acc += quant_e4m3(a) @ quant_e4m3(w * f8_scale).T / f8_scale
            continue
        acc += a @ w.T
    if bias is not None:
        acc += (-1.0 if bias_neg else 1.0) * np.asarray(bias, np.float64)[None, :N]
    crow = map_rows(cmap, B, Tn, J)
    if addend is not None:
        ar = map_rows(addmap, B, Tn, J)
        ad, _ = _gather(addend, ar, N)
        acc += ad
    okc = crow >= 0
    s1 = s2 = None
    if epi == EPI_BNRELU_BWD:
        x = np.zeros((M, N))
        x[okc] = X[crow[okc], :N]
        z = x * np.asarray(xscale, np.float64)[None, :N] + np.asarray(xshift, np.float64)[None, :N]
        acc = np.where(z > 0, acc, 0.0)
        if xdrop and drop is not None and drop[1] != 0:
            seed, thresh, inv_keep = drop
            e = crow[:, None].astype(np.int64) * _ld(X) + np.arange(N)[None, :]
            acc = acc * drop_mul(drop_key(seed, xsalt), thresh, inv_keep, np.where(okc[:, None], e, 0))
        if round_fn is not None:
            acc = round_fn(acc)
        s1, s2 = acc, acc * x
    elif epi == EPI_STATS:
        if round_fn is not None:
            acc = round_fn(acc)
        s1, s2 = acc, acc * acc
    C[crow[okc], :N] = acc[okc]
    if epi != EPI_PLAIN:
        nblk = (M + GEMM_BM - 1) // GEMM_BM
        for b in range(nblk):
            sl = slice(b * GEMM_BM, min(M, (b + 1) * GEMM_BM))
            msk = okc[sl, None]
            partials[b, :N, 0] = np.where(msk, s1[sl], 0).sum(axis=0)
            partials[b, :N, 1] = np.where(msk, s2[sl], 0).sum(axis=0)


def gemm_row_blocks(M):
    return (M + GEMM_BM - 1) // GEMM_BM


def wgrad(dom, P, R, pmap, segs, dW, drop=None, zero_first=True):
    """segs: list of dicts {Q, S, map, pro, scale, shift, salt, wcol0}.  dW fp32/64 [R][ldw] accumulated in place."""
    B, Tn, J = dom
    prow = map_rows(pmap, B, Tn, J)
    p, okp = _gather(P, prow, R)
    if zero_first:
        dW[:R, :] = 0
    for s in segs:
        rows = map_rows(s['map'], B, Tn, J)
        q, ok = _gather(s['Q'], rows, s['S'])
        q = _prologue(q, ok, rows, _ld(s['Q']), s['S'], s.get('pro', 0), s.get('scale'), s.get('shift'), s.get('salt', 0), drop)
        both = okp & ok
        dW[:R, s['wcol0']:s['wcol0'] + s['S']] += (p[both].T @ q[both])


# ------------------------------------------------------------------------------------------------ patterns
def build_pattern(adj_pattern):
    """int32 pattern array of gast_hip.h from a (J,J) 0/1 matrix (edges enumerated row-major, as the reference's
    boolean-mask assignment does, local_attention.py:41)."""
    m = np.asarray(adj_pattern) > 0
    J = m.shape[0]
    rows, cols = np.nonzero(m)  # row-major order
    nnz = rows.shape[0]
    row_ptr = np.zeros(J + 1, dtype=np.int32)
    for i in rows:
        row_ptr[i + 1] += 1
    row_ptr = np.cumsum(row_ptr).astype(np.int32)
    order = np.lexsort((rows, cols))  # by column then row
    col_ptr = np.zeros(J + 1, dtype=np.int32)
    for j in cols:
        col_ptr[j + 1] += 1
    col_ptr = np.cumsum(col_ptr).astype(np.int32)
    crow = rows[order].astype(np.int32)
    cedge = order.astype(np.int32)
    # padded fixed-degree (ELL) views (see include/gast_hip.h): padding slots carry edge id nnz (an all-zero weight row)
    Dr = int(np.diff(row_ptr).max())
    Dc = int(np.diff(col_ptr).max())
    ell_rj = np.repeat(np.arange(J), Dr).reshape(J, Dr)
    ell_rk = np.full((J, Dr), nnz)
    ell_ci = np.repeat(np.arange(J), Dc).reshape(J, Dc)
    ell_ck = np.full((J, Dc), nnz)
    for i in range(J):
        for d, k in enumerate(range(row_ptr[i], row_ptr[i + 1])):
            ell_rj[i, d], ell_rk[i, d] = cols[k], k
    for j in range(J):
        for d, q in enumerate(range(col_ptr[j], col_ptr[j + 1])):
            ell_ci[j, d], ell_ck[j, d] = crow[q], cedge[q]
    return np.concatenate([[J, nnz], row_ptr, cols.astype(np.int32), col_ptr, crow, cedge, [Dr, Dc], ell_rj.ravel(), ell_rk.ravel(),
                           ell_ci.ravel(), ell_ck.ravel()]).astype(np.int32)


def parse_pattern(pat):
    J, nnz = int(pat[0]), int(pat[1])
    o = 2
    row_ptr = pat[o:o + J + 1]; o += J + 1
    col = pat[o:o + nnz]; o += nnz
    col_ptr = pat[o:o + J + 1]; o += J + 1
    crow = pat[o:o + nnz]; o += nnz
    cedge = pat[o:o + nnz]
    erow = np.repeat(np.arange(J), np.diff(row_ptr))
    return J, nnz, row_ptr, col, col_ptr, crow, cedge, erow


# ------------------------------------------------------------------------------------------------ SemCH graph conv
def semch_adj_fwd(e, pat, A_t):
    J, nnz, row_ptr, col, _, _, _, erow = parse_pattern(pat)
    e = np.asarray(e, np.float64)
    for i in range(J):
        ks = slice(row_ptr[i], row_ptr[i + 1])
        z = e[:, ks] - e[:, ks].max(axis=1, keepdims=True)
        ex = np.exp(z)
        A_t[ks, :] = (ex / ex.sum(axis=1, keepdims=True)).T
    if A_t.shape[0] > nnz:
        A_t[nnz:, :] = 0      # the zero weight row the padded (ELL) edge slots point at


def semch_adj_bwd(dA_t, A_t, pat, de, accumulate=False):
    J, nnz, row_ptr, *_ = parse_pattern(pat)
    A = np.asarray(A_t, np.float64)
    dA = np.asarray(dA_t, np.float64)
    for i in range(J):
        ks = slice(row_ptr[i], row_ptr[i + 1])
        dot = (A[ks] * dA[ks]).sum(axis=0, keepdims=True)
        v = (A[ks] * (dA[ks] - dot)).T
        de[:, ks] = de[:, ks] + v if accumulate else v


def _agg_cfg(F, C):
    """(frames per block pass, frame blocks, channel chunks) of the forward aggregation: blocks of FB frames x 64 channels"""
    cc = min(C, 64)
    nchunk = (C + cc - 1) // cc
    fb = 256 // (cc // 4)
    nfb = min((F + fb - 1) // fb, max(1, 1024 // nchunk))
    return fb, nfb, nchunk


def _agg_frame_blocks(F, C):
    return _agg_cfg(F, C)[1]


def _agg_joint_split(F, C):
    """few frames: the rows (joints) of a frame are dealt to 4 (2) blocks; partial rows are then [joint part][frame block]"""
    _, nfb, nchunk = _agg_cfg(F, C)
    return 4 if nfb * nchunk <= 128 else 2 if nfb * nchunk <= 512 else 1


def semch_agg_blocks(F, C):
    return _agg_frame_blocks(F, C) * _agg_joint_split(F, C)


def semch_agg_fwd(H, F, J, C, A_sym, pat_sym, A_con, pat_con, Y, partials, round_fn=None, center_sym=None, center_con=None):
    Hf = np.asarray(H[:F * J, :4 * C], np.float64).reshape(F, J, 4 * C)
    out = np.zeros((F, J, 2 * C))
    for g, (A, pat) in enumerate(((A_sym, pat_sym), (A_con, pat_con))):
        _, nnz, row_ptr, col, _, _, _, erow = parse_pattern(pat)
        A = np.asarray(A, np.float64)
        h0 = Hf[:, :, g * 2 * C: g * 2 * C + C]
        h1 = Hf[:, :, g * 2 * C + C: g * 2 * C + 2 * C]
        for k in range(nnz):
            i, j = erow[k], col[k]
            src = h0 if i == j else h1
            out[:, i, g * C:(g + 1) * C] += A[k][None, :] * src[:, j, :]
    for g, ctr in enumerate((center_sym, center_con)):
        if ctr is not None:
            out[:, :, g * C:(g + 1) * C] -= np.asarray(ctr, np.float64)[None, None, :C]
    if round_fn is not None:
        out = round_fn(out)
    Y[:F * J, :2 * C] = out.reshape(F * J, 2 * C)
    # partial sums: frames are dealt round-robin to (block, slot): f = (it*nfb + blk)*FB + slot; joints i = part, part + split, ...
    nfb, split = _agg_frame_blocks(F, C), _agg_joint_split(F, C)
    fb = _agg_cfg(F, C)[0]
    partials[:nfb * split] = 0
    blk_of_frame = (np.arange(F) // fb) % nfb
    for part in range(split):
        for b in range(nfb):
            sel = out[blk_of_frame == b][:, part::split]
            partials[part * nfb + b, :2 * C, 0] = sel.sum(axis=(0, 1))
            partials[part * nfb + b, :2 * C, 1] = (sel * sel).sum(axis=(0, 1))


def semch_agg_bwd(dY, H, F, J, C, A_sym, pat_sym, A_con, pat_con, dH, dA_sym, dA_con, round_fn=None):
    """dA_* are accumulated (+=) like the kernel's atomics; the caller zeroes them."""
    Hf = np.asarray(H[:F * J, :4 * C], np.float64).reshape(F, J, 4 * C)
    dYf = np.asarray(dY[:F * J, :2 * C], np.float64).reshape(F, J, 2 * C)
    dHo = np.zeros((F, J, 4 * C))
    for g, (A, pat, dA) in enumerate(((A_sym, pat_sym, dA_sym), (A_con, pat_con, dA_con))):
        _, nnz, row_ptr, col, _, _, _, erow = parse_pattern(pat)
        A = np.asarray(A, np.float64)
        dy = dYf[:, :, g * C:(g + 1) * C]
        for k in range(nnz):
            i, j = erow[k], col[k]
            off = g * 2 * C + (0 if i == j else C)
            dHo[:, j, off:off + C] += A[k][None, :] * dy[:, i, :]
            dA[k, :] += (dy[:, i, :] * Hf[:, j, off:off + C]).sum(axis=0)
    if round_fn is not None:
        dHo = round_fn(dHo)
    dH[:F * J, :4 * C] = dHo.reshape(F * J, 4 * C)


# ------------------------------------------------------------------------------------------------ global attention
def _attn_common(AC, C_k, F, J, nheads):
    ac = np.asarray(AC[:F * J, :2 * nheads], np.float64).reshape(F, J, 2 * nheads)
    a = ac[:, :, :nheads].transpose(0, 2, 1)            # (F,h,J)  a_i
    c = ac[:, :, nheads:].transpose(0, 2, 1)            # (F,h,J)  c_j
    s = a[:, :, :, None] + c[:, :, None, :]             # (F,h,i,j)
    slope = np.where(s > 0, 1.0, 0.2)
    f = s * slope
    f = f - f.max(axis=-1, keepdims=True)
    ex = np.exp(f)
    p = ex / ex.sum(axis=-1, keepdims=True)
    att = p + np.asarray(C_k, np.float64).reshape(1, nheads, J, J)
    return p, att, slope


def attn_fwd(G, AC, C_k, F, J, C, nheads, Y, round_fn=None):
    Ci = C // nheads
    p, att, _ = _attn_common(AC, C_k, F, J, nheads)
    g = np.asarray(G[:F * J, :C], np.float64).reshape(F, J, nheads, Ci)
    y = np.einsum('fhij,fjhc->fihc', att, g).reshape(F * J, C)
    if round_fn is not None:
        y = round_fn(y)
    Y[:F * J, :C] = y


def attn_bwd(dY, G, AC, C_k, F, J, C, nheads, dG, dAC, dC_k, round_fn=None, dbias=None):
    """dC_k and dbias ([C + 2*nheads] = column sums of [dG | dAC]) are accumulated (+=); the caller zeroes them."""
    Ci = C // nheads
    p, att, slope = _attn_common(AC, C_k, F, J, nheads)
    g = np.asarray(G[:F * J, :C], np.float64).reshape(F, J, nheads, Ci)
    dy = np.asarray(dY[:F * J, :C], np.float64).reshape(F, J, nheads, Ci)
    datt = np.einsum('fihc,fjhc->fhij', dy, g)
    dg = np.einsum('fhij,fihc->fjhc', att, dy).reshape(F * J, C)
    dC_k += datt.sum(axis=0).reshape(dC_k.shape)
    ds = p * (datt - (p * datt).sum(axis=-1, keepdims=True)) * slope
    da = ds.sum(axis=3).transpose(0, 2, 1)              # (F,J,h)
    dc = ds.sum(axis=2).transpose(0, 2, 1)
    dac = np.concatenate([da, dc], axis=2).reshape(F * J, 2 * nheads)
    if dbias is not None:
        dbias[:C] += dg.sum(axis=0)        # column sums before storage rounding
        dbias[C:C + 2 * nheads] += dac.sum(axis=0)
    if round_fn is not None:
        dg, dac = round_fn(dg), round_fn(dac)
    dG[:F * J, :C] = dg
    dAC[:F * J, :2 * nheads] = dac


# ------------------------------------------------------------------------------------------------ BatchNorm pieces
def bn_finalize(partials, nblk, col0, N, count, gamma, beta, running_mean, running_var, nbt, momentum, eps,
                scale, shift, mean, rstd, centered=False):
    """centered: the statistics are those of x - running_mean (value before this call); see gast_hip.h."""
    p = np.asarray(partials[:nblk, col0:col0 + N], np.float64)
    s1, s2 = p[:, :, 0].sum(axis=0), p[:, :, 1].sum(axis=0)
    mu = s1 / count
    var = np.maximum(s2 / count - mu * mu, 0.0)
    r = 1.0 / np.sqrt(var + eps)
    sc = np.asarray(gamma, np.float64) * r
    scale[:N] = sc
    shift[:N] = np.asarray(beta, np.float64) - mu * sc
    mean[:N] = mu
    rstd[:N] = r
    if running_mean is not None:
        unb = var * (count / (count - 1.0)) if count > 1 else var
        running_mean[:N] = (running_mean[:N] + momentum * mu) if centered else ((1 - momentum) * running_mean[:N] + momentum * mu)
        running_var[:N] = (1 - momentum) * running_var[:N] + momentum * unb
    if nbt is not None:
        nbt[...] = nbt + 1


def bn_eval(gamma, beta, rm, rv, eps, N, scale, shift, centered=False):
    sc = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(rv, np.float64) + eps)
    scale[:N] = sc
    shift[:N] = np.asarray(beta, np.float64) - (0.0 if centered else np.asarray(rm, np.float64) * sc)


def bn_bwd_finalize(partials, nblk, col0, N, count, gamma, mean, rstd, dgamma, dbeta, ka, kb, kc, accumulate=False):
    p = np.asarray(partials[:nblk, col0:col0 + N], np.float64)
    s1, s2 = p[:, :, 0].sum(axis=0), p[:, :, 1].sum(axis=0)
    mu, r, g = (np.asarray(v, np.float64)[:N] for v in (mean, rstd, gamma))
    dg = r * (s2 - mu * s1)
    db = s1
    if accumulate:
        dgamma[:N] += dg
        dbeta[:N] += db
    else:
        dgamma[:N] = dg
        dbeta[:N] = db
    a = g * r
    b = -g * r * r * dg / count
    ka[:N] = a
    kb[:N] = b
    kc[:N] = -b * mu - a * db / count


def bn_bwd_apply(dz, X, rows, N, ka, kb, kc, round_fn=None):
    v = np.asarray(ka, np.float64)[:N] * dz[:rows, :N] + np.asarray(kb, np.float64)[:N] * X[:rows, :N] + np.asarray(kc, np.float64)[:N]
    dz[:rows, :N] = round_fn(v) if round_fn else v


def bn_bwd_apply_frames(dz, X, rows, N, ka, kb, kc, T_total, J, frames, round_fn=None):
    """gast_bn_bwd_apply_frames: rows = B * T_total * J; the rows of frames whose bit is clear in `frames` count as zero whatever they hold"""
    t = (np.arange(rows) // J) % T_total
    live = np.array([(int(frames) >> tt) & 1 for tt in range(T_total)], bool)[t]
    d = np.where(live[:, None], np.asarray(dz[:rows, :N], np.float64), 0.0)
    v = np.asarray(ka, np.float64)[:N] * d + np.asarray(kb, np.float64)[:N] * X[:rows, :N] + np.asarray(kc, np.float64)[:N]
    dz[:rows, :N] = round_fn(v) if round_fn else v


def bnrelu_apply(X, rows, N, scale, shift, Y, round_fn=None, use_drop=False, salt=0, drop=None):
    """drop = (seed, thresh, inv_keep); the stream is indexed by the element offset in X (row * ld(X) + col)."""
    v = np.maximum(np.asarray(X[:rows, :N], np.float64) * np.asarray(scale, np.float64)[:N] + np.asarray(shift, np.float64)[:N], 0)
    if use_drop and drop is not None and drop[1] != 0:
        seed, thresh, inv_keep = drop
        e = np.arange(rows, dtype=np.int64)[:, None] * _ld(X) + np.arange(N)[None, :]
        v = v * drop_mul(drop_key(seed, salt), thresh, inv_keep, e)
    Y[:rows, :N] = round_fn(v) if round_fn else v


def rowwise_blocks(rows, N):
    tpr = max(1, min(N // 4, 256))
    rb = 256 // tpr
    return min((rows + rb - 1) // rb, 1024)


def _rowwise_partials(vals_list, rows, N, partials):
    nblk = rowwise_blocks(rows, N)
    tpr = max(1, min(N // 4, 256))
    rb = 256 // tpr
    blk_of_row = (np.arange(rows) // rb) % nblk
    for b in range(nblk):
        sel = blk_of_row == b
        for q, v in enumerate(vals_list):
            partials[b, :N, q] = v[sel].sum(axis=0)


def bnrelu_bwd_mask(dY, X, rows, N, scale, shift, use_drop, salt, drop, dz, partials, round_fn=None):
    x = np.asarray(X[:rows, :N], np.float64)
    z = x * np.asarray(scale, np.float64)[:N] + np.asarray(shift, np.float64)[:N]
    d = np.where(z > 0, np.asarray(dY[:rows, :N], np.float64), 0.0)
    if use_drop and drop is not None and drop[1] != 0:
        seed, thresh, inv_keep = drop
        e = np.arange(rows)[:, None].astype(np.int64) * _ld(X) + np.arange(N)[None, :]
        d = d * drop_mul(drop_key(seed, salt), thresh, inv_keep, e)
    if round_fn is not None:
        d = round_fn(d)
    dz[:rows, :N] = d
    _rowwise_partials([d, d * x], rows, N, partials)


SHRINK_ROWS = 32      # rows per partial block of gast_shrink_bwd (csrc/norm_ops.hip)


def shrink_bwd_blocks(rows):
    return (rows + SHRINK_ROWS - 1) // SHRINK_ROWS


def shrink_fwd(O, rows, K, scale, shift, W, pred):
    """gast_shrink_fwd: reference gast_net.py:99,176-178 -- shrink = Conv2d(2C * 2^(L-1), 3, 1, bias=False) applied to
    relu(cat_bn(O)) of the last block: pred[r, d] = sum_k relu(scale[k] O[r, k] + shift[k]) W[d][k]"""
    z = np.maximum(np.asarray(O[:rows, :K], np.float64) * np.asarray(scale, np.float64)[:K] + np.asarray(shift, np.float64)[:K], 0.0)
    D = np.asarray(W).shape[0]
    pred[:rows, :D] = z @ np.asarray(W, np.float64)[:, :K].T


def shrink_bwd(dp, W, O, rows, K, scale, shift, dO, partials, round_fn=None):
    """gast_shrink_bwd: input gradient of the same through the ReLU of cat_bn (gast_net.py:176 backward) and the BatchNorm-backward
    column sums {sum dO, sum dO * O} of every SHRINK_ROWS-row block"""
    x = np.asarray(O[:rows, :K], np.float64)
    D = np.asarray(W).shape[0]
    g = np.asarray(dp[:rows, :D], np.float64) @ np.asarray(W, np.float64)[:, :K]
    d = np.where(x * np.asarray(scale, np.float64)[:K] + np.asarray(shift, np.float64)[:K] > 0, g, 0.0)
    if round_fn is not None:
        d = round_fn(d)
    dO[:rows, :K] = d
    for b in range(shrink_bwd_blocks(rows)):
        sl = slice(b * SHRINK_ROWS, min(rows, (b + 1) * SHRINK_ROWS))
        partials[b, :K, 0] = d[sl].sum(axis=0)
        partials[b, :K, 1] = (d[sl] * x[sl]).sum(axis=0)


def residual_fwd(O, omap, scO, shO, T2, sc2, sh2, use_drop, salt, drop, B, Tn, J, N, Xn, round_fn=None):
    rows = B * Tn * J
    orow = map_rows(omap, B, Tn, J)
    o, ok = _gather(O, orow, N)
    res = np.maximum(o * np.asarray(scO, np.float64)[:N] + np.asarray(shO, np.float64)[:N], 0.0)
    res[~ok] = 0
    t = np.maximum(np.asarray(T2[:rows, :N], np.float64) * np.asarray(sc2, np.float64)[:N] + np.asarray(sh2, np.float64)[:N], 0.0)
    if use_drop and drop is not None and drop[1] != 0:
        seed, thresh, inv_keep = drop
        e = np.arange(rows)[:, None].astype(np.int64) * _ld(T2) + np.arange(N)[None, :]
        t = t * drop_mul(drop_key(seed, salt), thresh, inv_keep, e)
    v = res + t
    Xn[:rows, :N] = round_fn(v) if round_fn else v


def colsum(X, rows, N, out, zero_first=True):
    if zero_first:
        out[:N] = 0
    out[:N] += np.asarray(X[:rows, :N], np.float64).sum(axis=0)


# ------------------------------------------------------------------------------------------------ input side
IN_ROWS_PER_BLOCK = 1024


def input_stats_blocks(rows):
    return (rows + IN_ROWS_PER_BLOCK - 1) // IN_ROWS_PER_BLOCK


def input_stats(x, rows, F_in, partials):
    xv = np.asarray(x, np.float64).reshape(rows, F_in)
    for b in range(input_stats_blocks(rows)):
        sl = xv[b * IN_ROWS_PER_BLOCK:(b + 1) * IN_ROWS_PER_BLOCK]
        partials[b, :F_in, 0] = sl.sum(axis=0)
        partials[b, :F_in, 1] = (sl * sl).sum(axis=0)


def _expand_taps(x, B, T_in, J, F_in, k0, t_stride):
    T_out = (T_in - k0) // t_stride + 1
    xv = np.asarray(x, np.float64).reshape(B, T_in, J, F_in)
    taps = np.stack([xv[:, tap: tap + (T_out - 1) * t_stride + 1: t_stride] for tap in range(k0)], axis=-1)  # (B,T_out,J,F,k0)
    return T_out, taps


def expand_fwd(x, B, T_in, J, F_in, k0, t_stride, W, sc0, sh0, C, E, partials, round_fn=None, center=None):
    T_out, taps = _expand_taps(x, B, T_in, J, F_in, k0, t_stride)
    xn = taps * np.asarray(sc0, np.float64)[None, None, None, :, None] + np.asarray(sh0, np.float64)[None, None, None, :, None]
    w = np.asarray(W, np.float64).reshape(C, F_in, k0)
    e = np.einsum('btjfk,cfk->btjc', xn, w).reshape(B * T_out * J, C)
    if center is not None:
        e = e - np.asarray(center, np.float64)[None, :C]
    if round_fn is not None:
        e = round_fn(e)
    rows = B * T_out * J
    E[:rows, :C] = e
    _rowwise_partials([e, e * e], rows, C, partials)


def expand_bwd(dE, x, B, T_in, J, F_in, k0, t_stride, mean0, rstd0, C, W, gamma0, beta0, dW, dgamma0, dbeta0, accumulate=False):
    """dW written (+= with accumulate), dgamma0 / dbeta0 accumulated (gast_hip.h)."""
    T_out, taps = _expand_taps(x, B, T_in, J, F_in, k0, t_stride)
    xh = (taps - np.asarray(mean0, np.float64)[None, None, None, :, None]) * np.asarray(rstd0, np.float64)[None, None, None, :, None]
    d = np.asarray(dE[:B * T_out * J, :C], np.float64).reshape(B, T_out, J, C)
    G = np.einsum('btjc,btjfk->cfk', d, xh)
    S = d.sum(axis=(0, 1, 2))
    w = np.asarray(W, np.float64).reshape(C, F_in, k0)
    g0 = np.asarray(gamma0, np.float64).reshape(1, F_in, 1)
    b0 = np.asarray(beta0, np.float64).reshape(1, F_in, 1)
    dWv = (g0 * G + b0 * S.reshape(C, 1, 1)).reshape(dW.shape)
    dW[...] = dW + dWv if accumulate else dWv
    dgamma0[...] = dgamma0 + (w * G).sum(axis=(0, 2))
    dbeta0[...] = dbeta0 + (w * S.reshape(C, 1, 1)).sum(axis=(0, 2))


def stream_shift(buf, newest):
    """gast_stream_shift_multi, one job: buf (B, Tb, X) advances by one frame in place (frame t <- frame t + 1) and takes newest (B, X)
    as its last frame -- the per-level frame windows of causal streaming inference (gast_hip/streaming.py)."""
    buf[:, :-1] = buf[:, 1:].copy()
    buf[:, -1] = newest


def prep(zero, seed=None, pad=None):
    """gast_prep (pass prologue): zero-fill every array of `zero`; seed = (counter, copy): copy[0] = counter[0] = counter[0] + 1
    (mod 2^32); pad = (src, dst, rows, cols_src, cols_dst[, scale]): dst[r, c] = scale * src[r, c] for c < cols_src, 0 beyond (the
    caller rounds dst to the 16-bit storage type where the destination has one)."""
    for z in zero:
        z[...] = 0
    if seed is not None:
        ctr, copy = seed
        v = (int(ctr.reshape(-1)[0]) + 1) & 0xffffffff
        if ctr.dtype == np.int32 and v >= 2 ** 31:
            v -= 2 ** 32
        ctr.reshape(-1)[0] = v
        copy.reshape(-1)[0] = v
    if pad is not None:
        src, dst, rows, cs, cd = pad[:5]
        d = dst.reshape(rows, cd)
        d[...] = 0
        d[:, :cs] = src.reshape(rows, cs) * (pad[5] if len(pad) > 5 else 1.0)

