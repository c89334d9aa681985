"""Host-side plan of the GAST-Net spatio-temporal forward/backward on the `gast_hip` op set.

The reference runs this path as ~900 ATen calls per step.  Here the same arithmetic is a short,
explicit list of fused launches over position-major `(B*T*J, C)` tensors:

  per GraphAttentionBlock (reference model/gast_net.py:22-33) with input X (P x C):
    G1   H = X . [W_sym0 | W_sym1 | W_con0 | W_con1 | W_g(4 heads) | v_theta | v_phi]^T + bias      one GEMM, N = 5C+8
    AGG  Y = [SemCH-aggregate(sym) | SemCH-aggregate(con)] (+ bn_1/bn_2 partial sums)          local_attention.py:40-48
    ATT  Ya = (softmax_j(leaky(a_i + c_j)) + C_k) . g                                          global_attention.py:60-78
    G2   Lpre = relu(bn_12(Y)) . W_lc^T          (+ stats)   local cat_conv/cat_bn             local_attention.py:142-143
    G3   Gpre = Ya . W_gc^T                      (+ stats)   global cat_conv/cat_bn            global_attention.py:122
    G4   Opre = [X | drop(relu(bn(Lpre))) | drop(relu(bn(Gpre)))] . W_bc^T  (+ stats)          gast_net.py:28-32
  per temporal level (gast_net.py:167-176 / :242-249) with input Opre (lazy BN+ReLU):
    T1pre = sum_tap relu(bn(Opre))[t*s + tap*d] . W_tap^T (+ stats);  T2pre = relu(bn(T1pre)) . W_1x1^T (+ stats)
    Xnext = relu(bn(Opre))[residual slice] + drop(relu(bn(T2pre)))
  BatchNorm is two-phase: producers emit partial column sums, `bn_finalize` makes scale/shift (and updates the running
  statistics), consumers apply BN+ReLU(+dropout) while loading their operand.  Nothing is permuted, concatenated or
  expanded in memory.

`ops` is the op set: `gast_hip.binding.HipOps` in the product (device tensors, HIP kernels, no fallback).  The test-suite
substitutes a numpy mirror of the same interface to check this file's composition on CPU against the reference's
golden fixtures (tests/fake_backend.py); that mirror lives in tests/ and is never importable from here.
"""
from collections import namedtuple

import contextlib
import os

import torch

RowMap = namedtuple('RowMap', 'T_total t_stride t_off')
MAX_SEG = 8     # K segments per gast_gemm / gast_wgrad launch (GAST_MAX_SEG, include/gast_hip.h; checked by tests/test_host_contract.py)
PRO_NONE, PRO_BNRELU, PRO_BNRELU_DROP = 0, 1, 2
EPI_PLAIN, EPI_STATS, EPI_BNRELU_BWD = 0, 1, 2
BN_MOMENTUM = 0.1
BN_EPS = 1e-5
NHEADS = 4


FUSED_BN_BWD_ROWS = 8192      # BatchNorm backward: finalize + apply in one launch up to this many rows


def ident(T):
    return RowMap(T, 1, 0)


class BNState:
    """scale/shift (+ mean/rstd in training) of one BatchNorm2d for the current batch."""
    __slots__ = ('scale', 'shift', 'mean', 'rstd', 'count')

    EVAL_COUNT = 1e300     # "infinitely many rows": the batch-statistics terms of the BatchNorm backward vanish

    def __init__(self, n, dev, count, pre=None):
        """pre = (scale, shift, mean, rstd) views of the eval-mode tables (already filled; mean / rstd None unless a backward pass
        may follow): nothing to compute for this state.  Eval-mode BatchNorm is the fixed affine map y = scale*x + shift, whose
        backward is dx = gamma*rstd*dz, dgamma = sum dz*(x-mean)*rstd, dbeta = sum dz with the RUNNING statistics -- exactly what
        the training-mode backward kernels compute for count -> infinity."""
        self.scale = pre[0] if pre is not None else torch.empty(n, dtype=torch.float32, device=dev)
        self.shift = pre[1] if pre is not None else torch.empty(n, dtype=torch.float32, device=dev)
        if pre is not None and pre[2] is not None:
            self.mean, self.rstd = pre[2], pre[3]
            self.count = self.EVAL_COUNT
        else:
            self.mean = torch.empty(n, dtype=torch.float32, device=dev)
            self.rstd = torch.empty(n, dtype=torch.float32, device=dev)
            self.count = count


class ZeroArena:
    """One zero-filled device buffer per pass, handed out in pieces.

    Split-K GEMMs, split-M weight gradients and column sums accumulate with atomics into zero-initialised destinations; zeroing
    each of them separately cost ~30 memset nodes (5 us apiece) per step.  The total is learnt on the first pass over a given
    input shape (individual allocations), afterwards ONE torch.zeros serves every request of the pass."""

    def __init__(self):
        self.totals = {}
        self.key = None
        self.buf = None
        self.off = 0

    def begin(self, key, dev, lazy=False):
        """lazy: the buffer is handed back UNZEROED and the caller zero-fills it as part of its pass prologue (ops.prep: one launch
        for this arena, the gradient buffers and the dropout-seed bump instead of a memset node each)."""
        self.key, self.dev, self.off = key, dev, 0
        n = self.totals.get(key)
        self.buf = (torch.empty if lazy else torch.zeros)(n, dtype=torch.uint8, device=dev) if n else None
        return self.buf if lazy else None

    def take(self, shape, dtype=torch.float32):
        nbytes = 1
        for d in shape:
            nbytes *= int(d)
        nbytes *= torch.empty((), dtype=dtype).element_size()
        span = (nbytes + 255) // 256 * 256
        off = self.off
        self.off += span
        if self.buf is None or off + span > self.buf.numel():
            self.buf = None                      # first pass over this shape (or a changed plan): count, allocate separately
            return torch.zeros(*shape, dtype=dtype, device=self.dev)
        return self.buf[off:off + nbytes].view(dtype).view(*shape)

    def end(self):
        self.totals[self.key] = self.off
        self.buf = None


class Engine:
    """Executes the plan for one model instance.  `spec` is built by model.gast_net (see `ModelSpec`)."""

    def __init__(self, spec, ops, centered=False):
        self.spec = spec
        self.ops = ops
        # centred storage (bf16 activations): every lazily-normalised pre-BN tensor is stored as x - running_mean so that its
        # bf16 rounding error scales with the spread of the channel, not with |mean|
        self.centered = bool(centered)
        self.za = ZeroArena()
        # (the HIP op set fuses the BatchNorm backward of dY / dE into their one reader; the numpy mirror of the tests has neither form)
        self.fuse_bn_bwd = bool(getattr(ops, 'fuses_bn_bwd', False))
        self.shrink_rowwise = hasattr(ops, 'shrink_fwd') and os.environ.get('GAST_SHRINK_KERNEL', '1') not in ('0', '')
        self._pre = {}           # eval mode: pre-filled (scale, shift) views per BNState name
        self._side = {}          # device -> side stream for independent branches of the plan
        self._keep = []          # operands of side-stream launches, kept alive until the join

    def _ctr(self, bn):
        return bn['running_mean'] if self.centered else None

    @staticmethod
    def loss_scale(dt):
        """GAST_HIP_DTYPE=f16: d loss / d prediction is multiplied by this power of two before the backward pass (every gradient is
        linear in it), so that the activation gradients -- 1e-7 .. 1e-3 at B = 128, i.e. subnormal or zero in binary16 -- are stored with
        full precision; the caller divides the parameter gradients by it again (model/gast_net.py).  GAST_F16_LOSS_SCALE overrides."""
        if dt != torch.float16:
            return 1.0
        v = float(os.environ.get('GAST_F16_LOSS_SCALE', '4096'))
        if not (v >= 1.0) or v != float(int(v)) or (int(v) & (int(v) - 1)):      # (2.5 or 3.0 are not powers of two)
            raise ValueError('GAST_F16_LOSS_SCALE must be a power of two >= 1')
        return v

    def _prep(self, arena_buf, prep, pad=None):
        """Pass prologue: the arena's zero fill + whatever the caller wants zeroed before this pass (flat gradient buffer, packed
        gradient scratch) + the dropout-seed bump (+ the padded copy of d loss / d pred) as ONE launch (gast_prep)."""
        prep = prep or {}
        zero = [t for t in [arena_buf] + list(prep.get('zero') or ()) if t is not None and t.numel()]
        seed = prep.get('seed')
        if zero or seed is not None or pad is not None:
            self.ops.prep(zero, seed=seed, pad=pad)

    # ------------------------------------------------------------------------------------------ helpers
    def _new(self, rows, cols, dt, dev, zero=False):
        return (torch.zeros if zero else torch.empty)(rows, cols, dtype=dt, device=dev)

    def _new_rows128(self, rows, cols, dt, dev):
        """(rows, cols) view of a buffer whose row pitch is a multiple of 128 bytes: the G1 output [h | g | a, c] is 5C + 8 columns
        wide, and with that pitch every row of every column block starts in the middle of a cache line for the aggregation /
        attention kernels that read it in 256-byte pieces."""
        per = 128 // torch.empty((), dtype=dt).element_size()
        return torch.empty(rows, (cols + per - 1) // per * per, dtype=dt, device=dev)[:, :cols]

    def _bn_forward(self, partials, nblk, col0, n, count, bn, st, training, off=0, centered=False):
        """bn: module-like with weight/bias/running_mean/running_var/num_batches_tracked; st: BNState (slice off..off+n)."""
        self._bn_forward_group([(partials, nblk, col0, n, count, bn, st, off)], training, centered=centered)

    def _eval_table(self, inp, bufs, dev, need_grad=False):
        """Eval mode: scale/shift of EVERY BatchNorm in one launch (they depend on parameters and running statistics only).
        Returns {state name: (scale, shift, mean, rstd)} views into one table; a state that spans two BatchNorms (bn_1|bn_2,
        lcat|gcat) gets their concatenation.  mean / rstd (running statistics) are only filled when a backward pass may follow
        (need_grad: frozen-BatchNorm fine-tuning, gradients of an eval-mode forward), else None."""
        sp = self.spec
        L = len(sp.fw)
        groups = [('bn0', ['init_bn']), ('bnE', ['expand_bn'])]
        for s in range(L):
            g = 'g%d.' % s
            if s > 0:
                groups += [('l%d.bn1' % s, ['l%d.bn0' % s]), ('l%d.bn2' % s, ['l%d.bn1' % s])]
            groups += [(g + 'bnY', [g + 'bn_1', g + 'bn_2']), (g + 'bnLG', [g + 'lcat_bn', g + 'gcat_bn']), (g + 'bnO', [g + 'cat_bn'])]
        total = sum((sum(inp[k + '.weight'].numel() for k in keys) + 3) // 4 * 4 for _, keys in groups)
        if need_grad and self.centered:
            # (an eval-mode forward with autograd merely enabled is common and must work; only an actual backward() is unsupported
            # with centred storage -- Engine.backward raises)
            need_grad = False
            self._no_eval_grad = True
        table = torch.empty(4 if need_grad else 2, total, dtype=torch.float32, device=dev)
        jobs, out, o = {}, {}, 0
        for name, keys in groups:
            o = (o + 3) // 4 * 4              # consumers load scale/shift with 16-byte accesses
            o0 = o
            for k in keys:
                n = inp[k + '.weight'].numel()
                eps = bufs[k].get('eps', BN_EPS)
                jobs.setdefault(eps, []).append((inp[k + '.weight'], inp[k + '.bias'], bufs[k]['running_mean'], bufs[k]['running_var'],
                                                 table[0, o:o + n], table[1, o:o + n],
                                                 self.centered and k != 'init_bn'))      # the network input is never stored centred
                if need_grad:     # parameter-sized torch ops, off the inference path
                    table[2, o:o + n] = bufs[k]['running_mean']
                    table[3, o:o + n] = torch.rsqrt(bufs[k]['running_var'].float() + eps)
                o += n
            out[name] = (table[0, o0:o], table[1, o0:o], table[2, o0:o] if need_grad else None, table[3, o0:o] if need_grad else None)
        for eps, js in jobs.items():          # one launch per distinct eps (the reference uses the default everywhere)
            self.ops.bn_eval_multi(js, eps)
        return out

    def _bn_forward_group(self, items, training, centered=False):
        """items: (partials, nblk, col0, n, count, bn, st, off) -- independent BatchNorms whose statistics are ready at the same
        point of the plan (bn_1 + bn_2, lcat_bn + gcat_bn): one multi-job finalize launch in training mode."""
        ops = self.ops
        if not training:
            if not self._pre:       # (eval states are normally pre-filled by _eval_table)
                for partials, nblk, col0, n, count, bn, st, off in items:
                    sl = slice(off, off + n)
                    ops.bn_eval(bn['weight'], bn['bias'], bn['running_mean'], bn['running_var'], bn.get('eps', BN_EPS), n, st.scale[sl],
                                st.shift[sl], centered=centered)
            return
        jobs = []
        for partials, nblk, col0, n, count, bn, st, off in items:
            sl = slice(off, off + n)
            jobs.append(dict(partials=partials, nblk=nblk, col0=col0, N=n, count=count, gamma=bn['weight'], beta=bn['bias'],
                             running_mean=bn['running_mean'], running_var=bn['running_var'], nbt=bn['num_batches_tracked'],
                             momentum=bn.get('momentum', BN_MOMENTUM), eps=bn.get('eps', BN_EPS), scale=st.scale[sl], shift=st.shift[sl], mean=st.mean[sl],
                             rstd=st.rstd[sl], centered=centered))
        ops.bn_finalize_multi(jobs)

    # ------------------------------------------------------------------------------------------ side stream
    # Opt-in (GAST_HIP_SIDE_STREAM=1): independent branches (the attention core of a block in the forward pass, the deferred
    # weight gradients in the backward pass) on a second stream; inside the captured hipGraph they become parallel branches.
    # Measured on MI355X: the branches do overlap (rocprofv3 shows concurrent kernels), but concurrent kernels slow each other
    # and the fork/join nodes add ~130 us of gaps per step: 3.36 ms with, 3.34 ms without -- so it is off by default.
    # Allocation discipline: every tensor a side-stream kernel touches is allocated on the main stream BEFORE the fork and is
    # kept alive (self._keep / locals) until AFTER the join, so the caching allocator never sees a cross-stream lifetime.
    def _fork(self, dev):
        if dev.type != 'cuda' or os.environ.get('GAST_HIP_SIDE_STREAM', '0') in ('0', ''):
            return None
        side = self._side.get(dev)
        if side is None:
            side = self._side[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        return side

    @staticmethod
    def _on(side):
        return torch.cuda.stream(side) if side is not None else contextlib.nullcontext()

    @staticmethod
    def _join(side):
        if side is not None:
            torch.cuda.current_stream(side.device).wait_stream(side)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x, inp, bufs, training, act_dtype, drop, need_grad=True, prep=None):
        """x: (B,T,J,F_in) fp32 contiguous device tensor.  inp: dict of packed fp32 tensors (see ModelSpec.pack).
        prep: optional dict(zero=[tensors to zero-fill before the pass], seed=(counter, per-pass copy) of the dropout stream) --
        merged with the arena's own zero fill into the pass prologue launch.
        bufs: dict of BN buffer dicts (+ momentum / eps of each module).  need_grad: a backward pass may follow (only matters in
        eval mode, where the BatchNorm states then also carry the running mean / rstd).
        Returns (pred (B,T',J,3) fp32, saved dict for backward)."""
        sp, ops = self.spec, self.ops
        dev = x.device
        B, T_in, J, F_in = x.shape
        L = len(sp.fw)
        dt = act_dtype
        sv = {'B': B, 'T_in': T_in, 'dt': dt, 'drop': drop, 'training': training}
        use_drop = training and drop is not None and drop.thresh != 0
        za = self.za
        self._prep(za.begin(('fwd', tuple(x.shape), dt, training), dev, lazy=True), prep)
        self._no_eval_grad = False
        self._pre = {} if training else self._eval_table(inp, bufs, dev, need_grad)
        sv['no_eval_grad'] = self._no_eval_grad
        pre = self._pre.get

        # ---- init_bn statistics + expand conv (gast_net.py:163-164)
        k0 = sp.fw[0]
        s0 = k0 if sp.strided else 1
        if T_in < k0:
            raise RuntimeError('input has %d frames, receptive field needs at least %d' % (T_in, sp.receptive_field))
        T = [(T_in - k0) // s0 + 1]
        rows_in = B * T_in * J
        bn0 = BNState(F_in, dev, rows_in, pre('bn0'))
        if training:
            nb = ops.input_stats_blocks(rows_in)
            part = torch.empty(nb, F_in, 2, dtype=torch.float32, device=dev)
            ops.input_stats(x, rows_in, F_in, part)
            self._bn_forward(part, nb, 0, F_in, rows_in, bufs['init_bn'] | inp_bn(inp, 'init_bn'), bn0, True)
        else:
            self._bn_forward(None, 0, 0, F_in, rows_in, bufs['init_bn'] | inp_bn(inp, 'init_bn'), bn0, False)
        C0 = sp.channels
        P0 = B * T[0] * J
        E = self._new(P0, C0, dt, dev)
        nbE = ops.rowwise_blocks(P0, C0)
        partE = torch.empty(nbE, C0, 2, dtype=torch.float32, device=dev)
        cen = self.centered
        ops.expand_fwd(x, B, T_in, J, F_in, k0, s0, inp['expand_w'], bn0.scale, bn0.shift, C0, E, partE,
                       center=self._ctr(bufs['expand_bn']))
        bnE = BNState(C0, dev, P0, pre('bnE'))
        self._bn_forward(partE, nbE, 0, C0, P0, bufs['expand_bn'] | inp_bn(inp, 'expand_bn'), bnE, training, centered=cen)
        # The first block's input X = relu(expand_bn(E)) is never materialised (round 5): its four readers -- G1, the X segment of G4 and
        # their two weight gradients -- apply BatchNorm + ReLU while loading E, like every other lazily-normalised tensor of the plan (the
        # prologue is a fused multiply-add + max the GEMM kernels execute either way).  One elementwise launch and 2 x P x C x s bytes
        # per step less.  GAST_LAZY_X0=0: the materialised form (bisecting aid).
        if os.environ.get('GAST_LAZY_X0', '1') not in ('0', ''):
            X, xpro = E, dict(pro=PRO_BNRELU, scale=bnE.scale, shift=bnE.shift)
        else:
            X, xpro = self._new(P0, C0, dt, dev), {}
            ops.bnrelu_apply(E, P0, C0, bnE.scale, bnE.shift, X)
        sv.update(x=x, bn0=bn0, E=E, bnE=bnE, T=T)

        # ---- masked-softmax adjacencies of every block (parameters only; local_attention.py:40-42): one launch for all of them
        adjs, jobs = [], []
        for s in range(L):
            Cs = C0 * (2 ** s)
            A_s = torch.empty(sp.nnz_sym + 1, Cs, dtype=torch.float32, device=dev)    # + the zero row padded edge slots point at
            A_c = torch.empty(sp.nnz_con + 1, Cs, dtype=torch.float32, device=dev)
            jobs += [(inp['g%d.e_sym' % s], sp.pat_sym(dev), A_s), (inp['g%d.e_con' % s], sp.pat_con(dev), A_c)]
            adjs.append((A_s, A_c))
        ops.semch_adj_fwd_multi(jobs)

        stages, levels = [], []
        for s in range(L):
            C = C0 * (2 ** s)
            if s > 0:
                # ---- temporal level s (gast_net.py:167-174 / :242-247): input = previous block's Opre (lazy BN+ReLU)
                prev = stages[-1]
                k = sp.kw[s]                    # taps: fw[s] (dilated / strided) or 2*pad+1 (dense=True ablation)
                Tp = T[-1]
                if sp.strided:
                    Tn = (Tp - k) // k + 1
                    taps = [RowMap(Tp, k, tap) for tap in range(k)]
                    resmap = RowMap(Tp, k, sp.causal_shift[s] + k // 2)
                else:
                    d = sp.tapstep[s]
                    Tn = Tp - (k - 1) * d
                    taps = [RowMap(Tp, 1, tap * d) for tap in range(k)]
                    resmap = RowMap(Tp, 1, sp.pad[s] + sp.causal_shift[s])
                if Tn < 1:
                    raise RuntimeError('input too short for the receptive field (%d frames needed)' % sp.receptive_field)
                T.append(Tn)
                P = B * Tn * J
                Wc = inp['l%d.conv' % s]        # [C][k*C], tap-major K (packed by gast_hip.packer)
                W1 = inp['l%d.conv1' % s]       # [C][C]
                nb = ops.gemm_row_blocks(P)
                T1 = self._new(P, C, dt, dev)
                part1 = za.take((nb, C, 2))
                segs = [dict(A=prev['O'], K=C, map=taps[tap], W=Wc[:, tap * C:(tap + 1) * C], pro=PRO_BNRELU,
                             scale=prev['bnO'].scale, shift=prev['bnO'].shift) for tap in range(k)]
                self._gemm_chunked((B, Tn, J), C, segs, T1, ident(Tn), epi=EPI_STATS, partials=part1,
                                   bias=self._ctr(bufs['l%d.bn0' % s]), bias_neg=cen)
                bn1 = BNState(C, dev, P, pre('l%d.bn1' % s))
                self._bn_forward(part1, nb, 0, C, P, bufs['l%d.bn0' % s] | inp_bn(inp, 'l%d.bn0' % s), bn1, training, centered=cen)
                T2 = self._new(P, C, dt, dev)
                part2 = za.take((nb, C, 2))
                ops.gemm((B, Tn, J), C, [dict(A=T1, K=C, map=ident(Tn), W=W1, pro=PRO_BNRELU, scale=bn1.scale, shift=bn1.shift)],
                         T2, ident(Tn), epi=EPI_STATS, partials=part2, bias=self._ctr(bufs['l%d.bn1' % s]), bias_neg=cen)
                bn2 = BNState(C, dev, P, pre('l%d.bn2' % s))
                self._bn_forward(part2, nb, 0, C, P, bufs['l%d.bn1' % s] | inp_bn(inp, 'l%d.bn1' % s), bn2, training, centered=cen)
                X = self._new(P, C, dt, dev)
                ops.residual_fwd(prev['O'], resmap, prev['bnO'].scale, prev['bnO'].shift, T2, bn2.scale, bn2.shift,
                                 use_drop, 3 * s, drop, B, Tn, J, C, X)
                levels.append(dict(T1=T1, T2=T2, bn1=bn1, bn2=bn2, taps=taps, resmap=resmap, k=k))
                xpro = {}
            stages.append(self._gab_forward(s, X, B, T[s], J, C, inp, bufs, training, dt, drop, use_drop, adjs[s], xpro=xpro))

        # ---- shrink (gast_net.py:99)
        last = stages[-1]
        CL = 2 * C0 * (2 ** (L - 1))
        PL = B * T[-1] * J
        Wsh = inp['shrink']   # [3][CL]
        pred = torch.empty(PL, 3, dtype=torch.float32, device=dev)
        if self.shrink_rowwise:
            # three output columns are a row-wise dot product, not a GEMM (round 6: 20 -> 6 us at B = 128; GAST_SHRINK_KERNEL=0: the GEMM)
            ops.shrink_fwd(last['O'], PL, CL, last['bnO'].scale, last['bnO'].shift, Wsh, pred)
        else:
            ops.gemm((B, T[-1], J), 3, [dict(A=last['O'], K=CL, map=ident(T[-1]), W=Wsh, pro=PRO_BNRELU,
                                             scale=last['bnO'].scale, shift=last['bnO'].shift)], pred, ident(T[-1]))
        sv.update(stages=stages, levels=levels)
        za.end()
        return pred.view(B, T[-1], J, 3), sv

    def _gab_forward(self, s, X, B, Tn, J, C, inp, bufs, training, dt, drop, use_drop, adj, xpro=None):
        """xpro: load prologue of X (BatchNorm + ReLU keys of a K segment) when X is a pre-BatchNorm tensor read lazily (block 0: E)."""
        xpro = xpro or {}
        sp, ops, za = self.spec, self.ops, self.za
        dev = X.device
        P = B * Tn * J
        F = B * Tn
        N1 = 5 * C + 2 * NHEADS
        g = 'g%d.' % s
        dom = (B, Tn, J)
        im = ident(Tn)
        Wg1 = inp[g + 'Bg1']      # [N1][C]
        Wlc = inp[g + 'Blc']      # [C][2C]
        Wgc = inp[g + 'Bgc']      # [C][C]
        Wbc = inp[g + 'Bbc']      # [2C][3C]
        nb = ops.gemm_row_blocks(P)
        # G1: everything that reads X in one pass (local_attention.py:37-38, global_attention.py:56-72)
        H = self._new_rows128(P, N1, dt, dev)
        ops.gemm(dom, N1, [dict(A=X, K=C, map=im, W=Wg1, **xpro)], H, im, bias=inp[g + 'bias1'])
        cen = self.centered
        # ---- everything both branches touch is allocated here, on the main stream, before the fork
        A_s, A_c = adj
        Y = self._new(P, 2 * C, dt, dev)
        nba = ops.semch_agg_blocks(F, C)
        partY = torch.empty(nba, 2 * C, 2, dtype=torch.float32, device=dev)
        bnY = BNState(2 * C, dev, P, self._pre.get(g + 'bnY'))
        # the local and the global branch write the two column halves of ONE tensor LG = [Lpre | Gpre] (and one BNState): their
        # post-activation drop(relu(bn(.))) is materialised once as ZLG, so the G4 GEMM / its weight gradient read a plain operand
        # (the dropout hash in their load prologues was 3/4 of the VALU stream of those kernels, 250 of 345 instructions per K tile)
        LG = self._new(P, 2 * C, dt, dev)
        Lp, Gp = LG[:, :C], LG[:, C:]
        partL = za.take((nb, C, 2))
        partG = za.take((nb, C, 2))
        bnLG = BNState(2 * C, dev, P, self._pre.get(g + 'bnLG'))
        Ya = self._new(P, C, dt, dev)
        ZLG = self._new(P, 2 * C, dt, dev)
        # ---- global branch (optionally on the side stream): attention core -> G3
        side = self._fork(dev)
        with self._on(side):
            ops.attn_fwd(H[:, 4 * C:5 * C], H[:, 5 * C:], inp[g + 'C_k'], F, J, C, NHEADS, Ya)
        # ---- local branch: neighbour aggregation + bn_1/bn_2 statistics (one finalize launch for both)
        ops.semch_agg_fwd(H, F, J, C, A_s, sp.pat_sym(dev), A_c, sp.pat_con(dev), Y, partY, deg=(sp.deg_sym[0], sp.deg_con[0]),
                          center=(self._ctr(bufs[g + 'bn_1']), self._ctr(bufs[g + 'bn_2'])))
        self._bn_forward_group([(partY, nba, 0, C, P, bufs[g + 'bn_1'] | inp_bn(inp, g + 'bn_1'), bnY, 0),
                                      (partY, nba, C, C, P, bufs[g + 'bn_2'] | inp_bn(inp, g + 'bn_2'), bnY, C)], training, centered=cen)
        self._join(side)
        # ---- G2 (local cat conv) and G3 (global cat conv) are independent: one grid, then one finalize for lcat_bn + gcat_bn
        ops.gemm_multi([dict(dom=dom, N=C, segs=[dict(A=Y, K=2 * C, map=im, W=Wlc, pro=PRO_BNRELU, scale=bnY.scale, shift=bnY.shift)],
                             C_=Lp, cmap=im, epi=EPI_STATS, partials=partL, bias=self._ctr(bufs[g + 'lcat_bn']), bias_neg=cen),
                        dict(dom=dom, N=C, segs=[dict(A=Ya, K=C, map=im, W=Wgc)], C_=Gp, cmap=im, epi=EPI_STATS, partials=partG,
                             bias=self._ctr(bufs[g + 'gcat_bn']), bias_neg=cen)])
        self._bn_forward_group([(partL, nb, 0, C, P, bufs[g + 'lcat_bn'] | inp_bn(inp, g + 'lcat_bn'), bnLG, 0),
                                       (partG, nb, 0, C, P, bufs[g + 'gcat_bn'] | inp_bn(inp, g + 'gcat_bn'), bnLG, C)], training, centered=cen)
        ops.bnrelu_apply(LG, P, 2 * C, bnLG.scale, bnLG.shift, ZLG, use_drop=use_drop, salt=3 * s + 1, drop=drop)
        # G4: cat(residual, local, global) . W (gast_net.py:28-32), concat never materialised: two K segments
        O = self._new(P, 2 * C, dt, dev)
        partO = za.take((nb, 2 * C, 2))
        segs = [dict(A=X, K=C, map=im, W=Wbc[:, 0:C], **xpro), dict(A=ZLG, K=2 * C, map=im, W=Wbc[:, C:3 * C])]
        ops.gemm(dom, 2 * C, segs, O, im, epi=EPI_STATS, partials=partO, bias=self._ctr(bufs[g + 'cat_bn']), bias_neg=cen)
        bnO = BNState(2 * C, dev, P, self._pre.get(g + 'bnO'))
        self._bn_forward(partO, nb, 0, 2 * C, P, bufs[g + 'cat_bn'] | inp_bn(inp, g + 'cat_bn'), bnO, training, centered=cen)
        return dict(X=X, H=H, A_s=A_s, A_c=A_c, Y=Y, bnY=bnY, Ya=Ya, LG=LG, ZLG=ZLG, bnLG=bnLG, Lp=Lp, Gp=Gp, O=O, bnO=bnO,
                    C=C, Tn=Tn, P=P, use_drop=use_drop, xpro=xpro)

    def _gemm_multi_chunked(self, jobs):
        """ops.gemm_multi for any number of independent jobs (GEMM_MAX_BATCH = 3 per launch)"""
        for i in range(0, len(jobs), 3):
            self.ops.gemm_multi(jobs[i:i + 3])

    def _gemm_chunked(self, dom, N, segs, out, cmap, addend=None, addmap=None, **epilogue):
        """ops.gemm for any number of K segments (the 7- and 19-tap convolutions of the dense=True ablation exceed the MAX_SEG
        segments of one launch): MAX_SEG segments per launch, each launch adding the previous partial result (`out` itself as
        the addend: every element is read and rewritten by the same thread), the caller's addend in the first and its epilogue
        (bias, statistics, ReLU/BN backward) in the last."""
        if len(segs) <= MAX_SEG:
            return self.ops.gemm(dom, N, segs, out, cmap, addend=addend, addmap=addmap, **epilogue)
        chunks = [segs[i:i + MAX_SEG] for i in range(0, len(segs), MAX_SEG)]
        for ci, ch in enumerate(chunks):
            a, am = (addend, addmap) if ci == 0 else (out, cmap)
            self.ops.gemm(dom, N, ch, out, cmap, addend=a, addmap=am, **(epilogue if ci == len(chunks) - 1 else {}))

    # ------------------------------------------------------------------------------------------ backward
    def _wgrad(self, dom, P, R, pmap, segs, dW, drop=None, zero_first=False):
        """Queue a weight gradient; the queue is flushed once per stage as ONE multi-job launch (ops.wgrad_multi).  Nothing in
        the backward pass reads a weight gradient, and every operand is in its final state when it is queued (the in-place
        BatchNorm backward of P runs before).  More than MAX_SEG segments (dense=True taps) become several jobs."""
        for i in range(0, len(segs), MAX_SEG):
            self._wq.append(dict(dom=dom, P=P, R=R, pmap=pmap, segs=segs[i:i + MAX_SEG], dW=dW, drop=drop,
                                 zero_first=zero_first and i == 0))

    def _wgrad_flush(self):
        """The queued weight gradients as one multi-job launch on the side stream: the next stage's backward chain (on the main
        stream) has ~35 launches that fill a handful of CUs each; the weight gradients soak up the rest of the machine."""
        if self._wq:
            side = self._fork(self._wq[0]['P'].device)
            with self._on(side):
                self.ops.wgrad_multi(self._wq)
            self._keep.append(self._wq)        # operands stay alive until the join at the end of backward()
            self._wside = side
            self._wq = []

    def _fin_flush(self):
        if self._finq:
            self.ops.rowsum_multi(self._finq)
            self._finq = []

    def _bn_backward(self, partials, nblk, col0, n, st, gamma, gout, key, dz, Xpre, rows, off=0, dzcol=None, frames=None):
        """finalize {sum dz, sum dz*x} -> dgamma/dbeta (ACCUMULATED into their destinations: the gradient buffers arrive zeroed or
        hold a running sum) + coefficients, then dz <- dx in place."""
        d = dz if dzcol is None else dz[:, dzcol:dzcol + n]
        xx = Xpre if dzcol is None else Xpre[:, dzcol:dzcol + n]
        self._bn_backward_group([dict(partials=partials, nblk=nblk, col0=col0, n=n, st=st, off=off, gamma=gamma, key=key, dz=d, X=xx,
                                      rows=rows, frames=frames)], gout)

    def _bn_backward_group(self, items, gout, one_apply=None):
        """items: dicts(partials, nblk, col0, n, st, off, gamma, key, dz, X, rows[, dzcol]) -- BatchNorm backward passes whose
        column sums are ready together: ONE multi-job finalize launch, then `dz <- dx` in place per item, or a single apply over
        adjacent column ranges of one tensor (one_apply = (dz, X, rows): bn_1 | bn_2)."""
        ops = self.ops
        dev = items[0]['gamma'].device
        ntot = sum(it['n'] for it in items)
        rows_max = one_apply[2] if one_apply is not None else max(it['rows'] for it in items)
        if rows_max <= FUSED_BN_BWD_ROWS:
            # short tensors (the M = B*J stage, small models): finalize + apply in ONE launch for the whole group
            jobs, o = [], 0
            for it in items:
                n, st, off = it['n'], it['st'], it.get('off', 0)
                sl = slice(off, off + n)
                if one_apply is not None:
                    dz, X, rows = one_apply[0][:, o:o + n], one_apply[1][:, o:o + n], one_apply[2]
                else:
                    dz, X, rows = it['dz'], it['X'], it['rows']
                jobs.append(dict(partials=it['partials'], nblk=it['nblk'], col0=it['col0'], N=n, count=st.count, gamma=it['gamma'],
                                 mean=st.mean[sl], rstd=st.rstd[sl], dgamma=gout[it['key'] + '.weight'],
                                 dbeta=gout[it['key'] + '.bias'], accumulate=True, dz=dz, X=X, rows=rows))
                o += n
            ops.bn_bwd_fused_multi(jobs)
            return
        ka = torch.empty(ntot, dtype=torch.float32, device=dev)
        kb = torch.empty(ntot, dtype=torch.float32, device=dev)
        kc = torch.empty(ntot, dtype=torch.float32, device=dev)
        jobs, o = [], 0
        for it in items:
            n, st, off = it['n'], it['st'], it.get('off', 0)
            sl = slice(off, off + n)
            jobs.append(dict(partials=it['partials'], nblk=it['nblk'], col0=it['col0'], N=n, count=st.count, gamma=it['gamma'],
                             mean=st.mean[sl], rstd=st.rstd[sl], dgamma=gout[it['key'] + '.weight'], dbeta=gout[it['key'] + '.bias'],
                             ka=ka[o:o + n], kb=kb[o:o + n], kc=kc[o:o + n], accumulate=True))
            o += n
        ops.bn_bwd_finalize_multi(jobs)
        if one_apply is not None:
            dz, X, rows = one_apply
            ops.bn_bwd_apply(dz, X, rows, ntot, ka, kb, kc)
            return
        o = 0
        for it in items:
            n = it['n']
            if it.get('frames') is not None:       # (T_total, J, bit mask): only these frames of dz were written, the rest counts as zero
                ops.bn_bwd_apply_frames(it['dz'], it['X'], it['rows'], n, ka[o:o + n], kb[o:o + n], kc[o:o + n], *it['frames'])
            else:
                ops.bn_bwd_apply(it['dz'], it['X'], it['rows'], n, ka[o:o + n], kb[o:o + n], kc[o:o + n])
            o += n

    def backward(self, sv, inp, dpred, gout, stage_done=None, prep=None):
        """dpred: (B,T',J,3) fp32.  Every gradient is written into its destination `gout[key]` (packed fp32 scratch
        regions for the GEMM operands, views of the flat gradient buffer for directly-held parameters).  The destinations must
        arrive ZERO-FILLED: split-M weight gradients, column sums and dC_k accumulate into them with atomics.
        stage_done(s), if given, is called for s = L-1 .. 1 as soon as every gradient of GraphAttentionBlock s has been enqueued
        (its weight gradients flushed, its adjacency-softmax backward run): the hook of the bucketed gradient exchange."""
        sp, ops = self.spec, self.ops
        dev = dpred.device
        if sv.get('no_eval_grad'):
            raise NotImplementedError('gradients of an eval-mode forward are not available with GAST_HIP_CENTER=1')
        B, dt, drop = sv['B'], sv['dt'], sv['drop']
        if dt != torch.float32 and hasattr(ops, 'set_h16'):
            ops.set_h16(dt)
        J = sp.J
        T = sv['T']
        L = len(sp.fw)
        C0 = sp.channels
        stages, levels = sv['stages'], sv['levels']
        grads = gout
        f32 = torch.float32
        za = self.za
        arena = za.begin(('bwd', B, sv['T_in'], dt), dev, lazy=True)
        self._wq = []
        self._wside = None
        self._adjq = []
        # the small reductions that end attn_bwd / semch_agg_bwd (partial rows -> dbias, dC_k, dA) are collected and run as ONE launch
        # in front of the adjacency-softmax backward, the only kernel that waits for them (6 launches per step -> 1)
        self._finq = [] if hasattr(ops, 'rowsum_multi') else None

        # ---- shrink backward
        last = stages[-1]
        CL = 2 * last['C']
        TL = T[-1]
        PL = B * TL * J
        KP = 8
        if dpred.dtype == f32 and dpred.is_contiguous() and (dt == f32 or getattr(ops, 'prep_pads_h16', False)):
            # written whole (3 columns + zero padding, the 16-bit modes' loss scale and rounding) by the pass prologue
            dp = torch.empty(PL, KP, dtype=dt, device=dev)
            self._prep(arena, prep, pad=(dpred, dp, PL, 3, KP, self.loss_scale(dt)))
        else:
            self._prep(arena, prep)
            dp = za.take((PL, KP), dt)
            ls = self.loss_scale(dt)
            dp[:, :3] = (dpred.reshape(PL, 3) * ls if ls != 1.0 else dpred.reshape(PL, 3)).to(dt)
        self._wgrad((B, TL, J), dp, KP, ident(TL), [dict(Q=last['O'], S=CL, map=ident(TL), pro=PRO_BNRELU, scale=last['bnO'].scale,
                                                        shift=last['bnO'].shift, wcol0=0)], gout['shrink'], zero_first=False)
        dO = self._new(PL, CL, dt, dev)
        if self.shrink_rowwise:
            nb = ops.shrink_bwd_blocks(PL)
            part = za.take((nb, CL, 2))
            ops.shrink_bwd(dp, inp['shrink'], last['O'], PL, CL, last['bnO'].scale, last['bnO'].shift, dO, part)
        else:
            WshT = inp['shrinkT']          # [CL][8], columns 3..7 zero
            nb = ops.gemm_row_blocks(PL)
            part = za.take((nb, CL, 2))
            ops.gemm((B, TL, J), CL, [dict(A=dp, K=KP, map=ident(TL), W=WshT)], dO, ident(TL), epi=EPI_BNRELU_BWD, partials=part,
                     X=last['O'], xscale=last['bnO'].scale, xshift=last['bnO'].shift)
        g = 'g%d.' % (L - 1)
        self._bn_backward(part, nb, 0, CL, last['bnO'], inp[g + 'cat_bn.weight'], grads, g + 'cat_bn', dO, last['O'], PL)

        # The input gradient dX of a block has two consumers: the residual path takes it as it is, the branch through
        # drop(relu(bn(T2pre))) (level s >= 1) or relu(expand_bn(E)) (the input side) takes it masked, with the BatchNorm-backward column
        # sums.  Fused (default): the block's last GEMM writes both -- masked value + sums through its BNRELU_BWD epilogue, the plain
        # value through the epilogue's second output (gast_gemm_args.C2) -- and the stand-alone gast_bnrelu_bwd_mask pass (3 launches,
        # 45 us per step at B = 128, a re-read of dX) is gone.  GAST_FUSE_MASK=0: the separate pass (bisecting aid).
        fuse_mask = os.environ.get('GAST_FUSE_MASK', '1') not in ('0', '')
        use_drop = sv['training'] and drop is not None and drop.thresh != 0
        P0 = B * T[0] * J
        for s in range(L - 1, -1, -1):
            st = stages[s]
            mask = None
            if fuse_mask:
                Pm, Cm = B * T[s] * J, st['C']
                mask = dict(dz=self._new(Pm, Cm, dt, dev), partials=za.take((ops.gemm_row_blocks(Pm), Cm, 2)), nblk=ops.gemm_row_blocks(Pm))
                if s > 0:
                    lv = levels[s - 1]
                    mask.update(X=lv['T2'], scale=lv['bn2'].scale, shift=lv['bn2'].shift, xdrop=use_drop, xsalt=3 * s, plain=True)
                else:
                    mask.update(X=sv['E'], scale=sv['bnE'].scale, shift=sv['bnE'].shift, xdrop=False, xsalt=0, plain=False)
            dX = self._gab_backward(s, st, dO, B, J, inp, grads, dt, drop, mask=mask)
            if s == 0:
                break
            # ---- temporal level s backward
            lv = levels[s - 1]
            prev = stages[s - 1]
            C = st['C']
            Tn, Tp = T[s], T[s - 1]
            P, Pp = B * Tn * J, B * Tp * J
            k = lv['k']
            # branch 2: drop(relu(bn(T2pre)))
            if mask is not None:
                dT2, part2, nbr = mask['dz'], mask['partials'], mask['nblk']
            else:
                nbr = ops.rowwise_blocks(P, C)
                part2 = torch.empty(nbr, C, 2, dtype=f32, device=dev)
                dT2 = self._new(P, C, dt, dev)
                ops.bnrelu_bwd_mask(dX, lv['T2'], P, C, lv['bn2'].scale, lv['bn2'].shift, use_drop, 3 * s, drop, dT2, part2)
            lk = 'l%d.' % s
            self._bn_backward(part2, nbr, 0, C, lv['bn2'], inp[lk + 'bn1.weight'], grads, lk + 'bn1', dT2, lv['T2'], P)
            # 1x1 conv
            self._wgrad((B, Tn, J), dT2, C, ident(Tn), [dict(Q=lv['T1'], S=C, map=ident(Tn), pro=PRO_BNRELU, scale=lv['bn1'].scale,
                                                            shift=lv['bn1'].shift, wcol0=0)], gout[lk + 'conv1'], zero_first=False)
            nbg = ops.gemm_row_blocks(P)
            part1 = za.take((nbg, C, 2))
            dT1 = self._new(P, C, dt, dev)
            W1T = inp[lk + 'conv1T']
            ops.gemm((B, Tn, J), C, [dict(A=dT2, K=C, map=ident(Tn), W=W1T)], dT1, ident(Tn), epi=EPI_BNRELU_BWD, partials=part1,
                     X=lv['T1'], xscale=lv['bn1'].scale, xshift=lv['bn1'].shift)
            self._bn_backward(part1, nbg, 0, C, lv['bn1'], inp[lk + 'bn0.weight'], grads, lk + 'bn0', dT1, lv['T1'], P)
            # temporal conv: weight gradient (k K-segments) ...
            self._wgrad((B, Tn, J), dT1, C, ident(Tn),
                      [dict(Q=prev['O'], S=C, map=lv['taps'][tap], pro=PRO_BNRELU, scale=prev['bnO'].scale, shift=prev['bnO'].shift,
                            wcol0=tap * C) for tap in range(k)], gout[lk + 'conv'], zero_first=False)
            # ... and input gradient, fused with the residual branch and the ReLU/BN backward of the previous block's output
            WcT = [inp[lk + 'convT'][tap * C:(tap + 1) * C] for tap in range(k)]     # [tap][cin][cout]
            pg = 'g%d.' % (s - 1)
            dO_frames = None
            if sp.strided:
                covered = k * Tn == Tp
                dOp = self._new(Pp, C, dt, dev) if covered else za.take((Pp, C), dt)
                nbt = ops.gemm_row_blocks(P)
                partO = za.take((k * nbt, C, 2))
                res_tap = lv['resmap'].t_off
                # (the k taps write disjoint rows of dOp: independent jobs of one grid -- and one split-K finish on the M = B*J stage)
                self._gemm_multi_chunked([dict(dom=(B, Tn, J), N=C, segs=[dict(A=dT1, K=C, map=ident(Tn), W=WcT[tap])], C_=dOp,
                                               cmap=RowMap(Tp, k, tap), addend=dX if tap == res_tap else None,
                                               addmap=ident(Tn) if tap == res_tap else None, epi=EPI_BNRELU_BWD,
                                               partials=partO[tap * nbt:(tap + 1) * nbt], X=prev['O'], xscale=prev['bnO'].scale,
                                               xshift=prev['bnO'].shift) for tap in range(k)])
                nbo = k * nbt
            elif Tn <= sp.tapstep[s] and any(tap * sp.tapstep[s] == lv['resmap'].t_off for tap in range(k)):
                # disjoint taps (few output frames: Tn <= dilation, e.g. the last level where Tn = 1): every input frame receives
                # at most ONE tap, so the input gradient is k scatter GEMMs over the OUTPUT domain instead of a gather GEMM over
                # the input domain whose rows are mostly zero (arc 3,3,3, last level: 3 x 1.1 GFLOP instead of 65 GFLOP).  Frames
                # no tap reaches keep the zero of the arena: their gradient before the BatchNorm backward is exactly zero.
                d = sp.tapstep[s]
                # (round 6) ... unless the BatchNorm-backward apply can be told which frames were written: then nothing is zero-filled
                # and nothing zero is read back (84.7 MB each way at B = 128)
                sparse = hasattr(ops, 'bn_bwd_apply_frames') and Tp <= 64 and Pp > FUSED_BN_BWD_ROWS and \
                    os.environ.get('GAST_SPARSE_TAP_GRAD', '1') not in ('0', '')
                dO_frames = (Tp, J, sum(1 << (tap * d + t) for tap in range(k) for t in range(Tn))) if sparse else None
                dOp = self._new(Pp, C, dt, dev) if sparse else za.take((Pp, C), dt)
                nbt = ops.gemm_row_blocks(P)
                partO = za.take((k * nbt, C, 2))
                res_off = lv['resmap'].t_off
                self._gemm_multi_chunked([dict(dom=(B, Tn, J), N=C, segs=[dict(A=dT1, K=C, map=ident(Tn), W=WcT[tap])], C_=dOp,
                                               cmap=RowMap(Tp, 1, tap * d), addend=dX if tap * d == res_off else None,
                                               addmap=ident(Tn) if tap * d == res_off else None, epi=EPI_BNRELU_BWD,
                                               partials=partO[tap * nbt:(tap + 1) * nbt], X=prev['O'], xscale=prev['bnO'].scale,
                                               xshift=prev['bnO'].shift) for tap in range(k)])
                nbo = k * nbt
            else:
                d = sp.tapstep[s]
                dOp = self._new(Pp, C, dt, dev)
                nbo = ops.gemm_row_blocks(Pp)
                partO = za.take((nbo, C, 2))
                segs = [dict(A=dT1, K=C, map=RowMap(Tn, 1, -tap * d), W=WcT[tap]) for tap in range(k)]
                self._gemm_chunked((B, Tp, J), C, segs, dOp, ident(Tp), addend=dX, addmap=RowMap(Tn, 1, -lv['resmap'].t_off),
                                   epi=EPI_BNRELU_BWD, partials=partO, X=prev['O'], xscale=prev['bnO'].scale,
                                   xshift=prev['bnO'].shift)
            self._bn_backward(partO, nbo, 0, C, prev['bnO'], inp[pg + 'cat_bn.weight'], grads, pg + 'cat_bn', dOp, prev['O'], Pp,
                              frames=dO_frames)
            dO = dOp
            self._wgrad_flush()
            if stage_done is not None:
                self._fin_flush()
                ops.semch_adj_bwd_multi(self._adjq, accumulate=True)     # (otherwise: one launch for all blocks at the end)
                self._adjq = []
                self._join(self._wside)
                stage_done(s)

        # ---- expand conv + init_bn backward (dX is the gradient w.r.t. relu(expand_bn(E)))
        if mask is not None:
            dE, partE, nbr = mask['dz'], mask['partials'], mask['nblk']
        else:
            nbr = ops.rowwise_blocks(P0, C0)
            partE = torch.empty(nbr, C0, 2, dtype=f32, device=dev)
            dE = self._new(P0, C0, dt, dev)
            ops.bnrelu_bwd_mask(dX, sv['E'], P0, C0, sv['bnE'].scale, sv['bnE'].shift, False, 0, None, dE, partE)
        # dE has ONE reader, the expand-conv backward: it applies the backward of expand_bn while it loads dE
        # instead of a stand-alone apply pass (GAST_FUSE_EXPAND_BN=0: off)
        bn_exp = {}
        if P0 > FUSED_BN_BWD_ROWS and self.fuse_bn_bwd and os.environ.get('GAST_FUSE_EXPAND_BN', '1') not in ('0', ''):
            kabc = torch.empty(3, C0, dtype=f32, device=dev)
            bnE = sv['bnE']
            job = dict(partials=partE, nblk=nbr, col0=0, N=C0, count=bnE.count, gamma=inp['expand_bn.weight'], mean=bnE.mean, rstd=bnE.rstd,
                       dgamma=grads['expand_bn.weight'], dbeta=grads['expand_bn.bias'], ka=kabc[0], kb=kabc[1], kc=kabc[2], accumulate=True)
            bn_exp = dict(bn=(sv['E'], kabc[0], kabc[1], kabc[2]))
            ops.bn_bwd_finalize_multi([job])
        else:
            self._bn_backward(partE, nbr, 0, C0, sv['bnE'], inp['expand_bn.weight'], grads, 'expand_bn', dE, sv['E'], P0)
        x = sv['x']
        F_in = x.shape[-1]
        k0 = sp.fw[0]
        s0 = k0 if sp.strided else 1
        # G / S sums and the parameter-sized epilogue (xn = gamma0*xhat + beta0 feeds the expand conv) in two launches
        ops.expand_bwd(dE, x, B, sv['T_in'], J, F_in, k0, s0, sv['bn0'].mean, sv['bn0'].rstd, C0, inp['expand_w'],
                       inp['init_bn.weight'], inp['init_bn.bias'], gout['expand_w'], gout['init_bn.weight'], gout['init_bn.bias'],
                       accumulate=True, **bn_exp)
        self._wgrad_flush()
        self._fin_flush()
        ops.semch_adj_bwd_multi(self._adjq, accumulate=True)
        self._adjq = []
        self._join(self._wside)
        self._keep = []
        za.end()

    def _gab_backward(self, s, st, dO, B, J, inp, grads, dt, drop, mask=None):
        """dO: gradient w.r.t. Opre (pre-BN output of the block's cat_conv), (P x 2C).  Returns dX (P x C).
        mask (see Engine.backward): the block's last GEMM also produces the masked gradient of the NEXT consumer's BatchNorm'd branch
        (mask['dz'], with its column sums in mask['partials']); dX itself is then only written when mask['plain']."""
        sp, ops, za = self.spec, self.ops, self.za
        dev = dO.device
        f32 = torch.float32
        C, Tn, P = st['C'], st['Tn'], st['P']
        F = B * Tn
        N1 = 5 * C + 2 * NHEADS
        g = 'g%d.' % s
        dom = (B, Tn, J)
        im = ident(Tn)
        nb = ops.gemm_row_blocks(P)
        xdrop = st['use_drop']
        # G4 weight gradient: two K segments (X | ZLG), plain operands
        xpro = st.get('xpro') or {}      # (block 0: X is E read through expand_bn + ReLU)
        self._wgrad(dom, dO, 2 * C, im, [dict(Q=st['X'], S=C, map=im, wcol0=0, **xpro), dict(Q=st['ZLG'], S=2 * C, map=im, wcol0=C)],
                    grads[g + 'Bbc'], zero_first=False)
        WbcT = inp[g + 'BbcT']       # [3C][2C]
        # input gradient of the local | global branches: ONE GEMM onto the column halves of dLG, fused with ReLU + dropout (the
        # mask is re-derived from the pre-BN tensor LG and the dropout stream of the forward) + the BN-sum backward; one finalize
        # launch for lcat_bn + gcat_bn and one in-place apply over both halves
        dLG = self._new(P, 2 * C, dt, dev)
        dL, dG = dLG[:, :C], dLG[:, C:]
        partLG = za.take((nb, 2 * C, 2))
        bnLG = st['bnLG']
        ops.gemm(dom, 2 * C, [dict(A=dO, K=2 * C, map=im, W=WbcT[C:3 * C])], dLG, im, epi=EPI_BNRELU_BWD, partials=partLG, X=st['LG'],
                 xscale=bnLG.scale, xshift=bnLG.shift, xdrop=xdrop, xsalt=3 * s + 1, drop=drop)
        self._bn_backward_group([dict(partials=partLG, nblk=nb, col0=0, n=C, st=bnLG, off=0, gamma=inp[g + 'lcat_bn.weight'],
                                      key=g + 'lcat_bn'),
                                 dict(partials=partLG, nblk=nb, col0=C, n=C, st=bnLG, off=C, gamma=inp[g + 'gcat_bn.weight'],
                                      key=g + 'gcat_bn')], grads, one_apply=(dLG, st['LG'], P))
        # local / global cat conv: weight gradients (queued) and input gradients (independent: one grid)
        self._wgrad(dom, dL, C, im, [dict(Q=st['Y'], S=2 * C, map=im, pro=PRO_BNRELU, scale=st['bnY'].scale, shift=st['bnY'].shift,
                                          wcol0=0)], grads[g + 'Blc'], zero_first=False)
        self._wgrad(dom, dG, C, im, [dict(Q=st['Ya'], S=C, map=im, wcol0=0)], grads[g + 'Bgc'], zero_first=False)
        WlcT = inp[g + 'BlcT']       # [2C][C]
        WgcT = inp[g + 'BgcT']
        dY = self._new(P, 2 * C, dt, dev)
        partY = za.take((nb, 2 * C, 2))
        dYa = self._new(P, C, dt, dev)
        ops.gemm_multi([dict(dom=dom, N=2 * C, segs=[dict(A=dL, K=C, map=im, W=WlcT)], C_=dY, cmap=im, epi=EPI_BNRELU_BWD,
                             partials=partY, X=st['Y'], xscale=st['bnY'].scale, xshift=st['bnY'].shift),
                        dict(dom=dom, N=C, segs=[dict(A=dG, K=C, map=im, W=WgcT)], C_=dYa, cmap=im)])
        itemsY = [dict(partials=partY, nblk=nb, col0=0, n=C, st=st['bnY'], off=0, gamma=inp[g + 'bn_1.weight'], key=g + 'bn_1'),
                  dict(partials=partY, nblk=nb, col0=C, n=C, st=st['bnY'], off=C, gamma=inp[g + 'bn_2.weight'], key=g + 'bn_2')]
        # dY has ONE reader, the aggregation backward: where its kernel can, it applies the BatchNorm backward of bn_1 | bn_2 while it
        # stages dY -- the stand-alone apply pass over P x 2C disappears (GAST_FUSE_AGG_BN=0: off)
        fuse_bn = (P > FUSED_BN_BWD_ROWS and self.fuse_bn_bwd and hasattr(ops, 'semch_agg_bwd_fuses_bn')
                   and os.environ.get('GAST_FUSE_AGG_BN', '1') not in ('0', '')
                   and ops.semch_agg_bwd_fuses_bn(st['H'], F, J, C, st['A_s'], st['A_c'], (sp.deg_sym[1], sp.deg_con[1])))
        bn_agg = None
        if fuse_bn:
            kabc = torch.empty(3, 2 * C, dtype=f32, device=dev)
            jobs, o = [], 0
            for it in itemsY:
                sl = slice(it['off'], it['off'] + C)
                jobs.append(dict(partials=partY, nblk=nb, col0=it['col0'], N=C, count=st['bnY'].count, gamma=it['gamma'], mean=st['bnY'].mean[sl],
                                 rstd=st['bnY'].rstd[sl], dgamma=grads[it['key'] + '.weight'], dbeta=grads[it['key'] + '.bias'],
                                 ka=kabc[0, o:o + C], kb=kabc[1, o:o + C], kc=kabc[2, o:o + C], accumulate=True))
                o += C
            bn_agg = dict(bn=(st['Y'], kabc[0], kabc[1], kabc[2]))
            ops.bn_bwd_finalize_multi(jobs)
        else:
            self._bn_backward_group(itemsY, grads, one_apply=(dY, st['Y'], P))
        # attention core + aggregation backward fill the column blocks of dH
        H = st['H']
        dH = self._new_rows128(P, N1, dt, dev)
        dCk = grads[g + 'C_k']          # accumulated with atomics: gradient destinations arrive zeroed (see backward())
        # (+ the bias gradients of g / theta / phi = column sums of these dH columns, reduced in the same pass)
        dkw = {'defer': self._finq} if self._finq is not None else {}
        ops.attn_bwd(dYa, H[:, 4 * C:5 * C], H[:, 5 * C:], inp[g + 'C_k'], F, J, C, NHEADS, dH[:, 4 * C:5 * C], dH[:, 5 * C:], dCk,
                     dbias=grads[g + 'bias1'][4 * C:], **dkw)
        nnz_s, nnz_c = sp.nnz_sym, sp.nnz_con
        dA = torch.empty(nnz_s + nnz_c, C, dtype=f32, device=dev)
        ws = torch.empty(max(1, ops.semch_agg_bwd_ws(F, C, nnz_s, nnz_c)), dtype=f32, device=dev)
        ops.semch_agg_bwd(dY, H, F, J, C, st['A_s'], sp.pat_sym(dev), st['A_c'], sp.pat_con(dev), dH, dA, ws,
                          cdeg=(sp.deg_sym[1], sp.deg_con[1]), **dkw, **(bn_agg or {}))
        # softmax backward of the adjacencies: queued, one launch for all blocks at the end of backward()
        self._adjq += [(dA[:nnz_s], st['A_s'], sp.pat_sym(dev), grads[g + 'e_sym']),
                       (dA[nnz_s:], st['A_c'], sp.pat_con(dev), grads[g + 'e_con'])]
        # G1 backward: one fat weight-gradient and one fat input-gradient GEMM
        self._wgrad(dom, dH, N1, im, [dict(Q=st['X'], S=C, map=im, wcol0=0, **xpro)], grads[g + 'Bg1'], zero_first=False)
        Wg1T = inp[g + 'Bg1T']       # [C][N1]
        # (the input side, s = 0 with the fused mask: nobody reads the plain value -- no buffer)
        dX = self._new(P, C, dt, dev) if (mask is None or mask['plain']) else None
        segs = [dict(A=dH, K=N1, map=im, W=Wg1T), dict(A=dO, K=2 * C, map=im, W=WbcT[0:C])]
        if mask is None:
            ops.gemm(dom, C, segs, dX, im)
        else:
            ops.gemm(dom, C, segs, mask['dz'], im, epi=EPI_BNRELU_BWD, partials=mask['partials'], X=mask['X'], xscale=mask['scale'],
                     xshift=mask['shift'], xdrop=mask['xdrop'], xsalt=mask['xsalt'], drop=drop, C2=dX if mask['plain'] else None)
        return dX


def inp_bn(inp, key):
    return {'weight': inp[key + '.weight'], 'bias': inp[key + '.bias']}
