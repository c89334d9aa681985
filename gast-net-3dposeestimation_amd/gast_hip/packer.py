"""Parameter packing / gradient unpacking tables for one model instance.

The state_dict contract fixes 165 parameter tensors in PyTorch layouts.  The plan wants a handful of
`[N][K]` K-contiguous GEMM operands in the activation dtype, in both orientations, with the attention's theta/phi folded
(see csrc/pack_ops.hip).  `Packer` describes that mapping ONCE as three job lists

  copy jobs    strided 2-D copies  parameter -> packed operand region (+ its transposed twin)
  fold jobs    v_theta/v_phi/a_theta/a_phi of every attention head
  unpack jobs  packed weight gradient (fp32 scratch) -> parameter-shaped gradient; unfold jobs for theta/phi/concat_project

and runs each list with ONE launch (`ops.run_copy/run_fold/run_unfold`).  Addresses in the tables are (base id, byte offset)
pairs, so the tables survive buffers that move between calls (the gradient buffer).  The numpy mirror used by the CPU tests
interprets the same python-level job descriptions through tensor views instead of raw addresses.
"""
import os

import torch

NHEADS = 4
BASE_ABS, BASE_W, BASE_F, BASE_S, BASE_G = 0, 1, 2, 3, 4   # absolute, packed weights (act dtype), packed fp32, grad scratch, grad buffer


class Ref:
    """A strided 2-D view: base (a tensor for BASE_ABS, else a base id), element offset, element strides."""
    __slots__ = ('base', 'tensor', 'off', 'rs', 'cs')

    def __init__(self, base, tensor, off, rs, cs):
        self.base, self.tensor, self.off, self.rs, self.cs = base, tensor, int(off), int(rs), int(cs)


class Layout:
    """Named 2-D regions inside one flat buffer (offsets in elements, 16-byte aligned)."""

    def __init__(self, align=8):
        self.regions = {}
        self.size = 0
        self.align = align

    def add(self, name, rows, cols):
        self.regions[name] = (self.size, rows, cols)
        self.size += (rows * cols + self.align - 1) // self.align * self.align

    def view(self, buf, name):
        off, r, c = self.regions[name]
        return buf[off:off + r * c].view(r, c)

    def ref(self, base, name, row0=0, col0=0, transposed=False):
        off, r, c = self.regions[name]
        if transposed:      # element (i, j) of the logical block lands at [col0 + j... ] of the region -> swap strides
            return Ref(base, None, off + row0 * c + col0, 1, c)
        return Ref(base, None, off + row0 * c + col0, c, 1)


# Raw writers of parameter memory that bypass torch's version counters (FlatAdam's HIP kernel writes the flat buffer the parameters
# are views of) bump this epoch; together with the parameters' `_version`s it tells `Packer.unchanged()` whether the packed operands
# still match the parameters -- an inference / evaluation loop then skips the three packing launches (95 us per forward at C = 128).
PARAM_EPOCH = [0]
# ... but a raw writer that was CAPTURED into a hipGraph (the whole training step of bench.py / a user's torch.cuda.graph) writes the
# parameters on every replay with no host-side trace at all: from then on the skip is off for good.
CAPTURED_WRITER = [False]


def note_raw_parameter_write():
    PARAM_EPOCH[0] += 1
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        CAPTURED_WRITER[0] = True


X3_PAD_ROWS = 256      # zero rows behind every pre-split weight image (a GEMM tile spans 256 weight rows from any start row)


def x3_image_rows(R):
    return (R + 15) // 16 * 16 + X3_PAD_ROWS


def x3_forward_f16():
    """GAST_X3_FWD = f16 (default) | bf16: the operand pairs of the FORWARD GEMMs in GAST_HIP_DTYPE=bf16x3.  f16: fp16 hi/lo pairs
    (GAST_F32X3H, include/gast_hip.h -- 22 instead of 16 significand bits at the same matrix-core rate; post-BatchNorm activations
    and weights live well inside fp16's range); input and weight gradients always run on bf16 pairs (fp32's exponent range)."""
    v = os.environ.get('GAST_X3_FWD', 'f16').lower()
    if v not in ('f16', 'fp16', 'bf16'):
        raise ValueError('GAST_X3_FWD must be f16 or bf16, got %r' % v)
    return v != 'bf16'


def h16_images(dt=None):
    """GAST_H16_IMAGES = 1 (default) | 0: the 16-bit modes keep a k-group-major layout image of every packed operand, so that their
    large-M GEMMs take gemm_big.hip (round 5); 0 = the round-1..4 behaviour (every GEMM on the 128 x 128 kernel).  Round 6 measured
    the switch again on four boxes (profiles/r06_ab_f16_kernels.txt): bfloat16 storage is faster WITH images (2.469 vs 2.503 ms),
    binary16 20 - 40 us faster WITHOUT (2.521 vs 2.543, 2.530 vs 2.568, 2.604 vs 2.627) -- but without them configs[2] moves from
    9.66e-3 to 1.0009e-2 against the float64 oracle, across the north star's 1e-2 bound (the 128 x 128 kernel sums a K tile in a
    different order).  Parity first: the images stay on for both storage types."""
    return os.environ.get('GAST_H16_IMAGES', '1') not in ('0', '')


class X3Weight:
    """An fp32 [N][K] GEMM operand together with its pre-split image (GAST_F32X3 / GAST_F32X3H; `gast_x3_image_multi`,
    include/gast_hip.h): img is the k-group-major 3-D view [ceil(K/16)][rows + zero padding][32].  f16: the GEMMs that read this
    operand run on fp16 pairs (and the image, when present, holds fp16 halves) -- set on the forward operands, never on the
    transposed twins the input gradients read.  Slices like the tensor it wraps (`w[r0:r1]`, `w[:, k0:k1]`); a column slice that is
    not aligned to the 16-value groups drops the image (the GEMM then splits the fp32 operand itself).
    group = 32 (round 5): a 16-BIT operand with its k-group-major LAYOUT image (32 values per 64-byte row, one product: the large-M
    kernel in the 16-bit modes, `gast_x3_image_job.f16 == 2`); f16 is False there -- the element type is the build's storage type."""
    __slots__ = ('t', 'img', 'f16', 'group')

    def __init__(self, t, img, f16=False, group=16):
        self.t, self.img, self.f16, self.group = t, img, bool(f16), int(group)

    def __getitem__(self, idx):
        rs, cs = idx if isinstance(idx, tuple) else (idx, slice(None))
        img = None
        if self.img is not None:
            R, K = self.t.shape
            a, b, step = cs.indices(K)
            r0, _, rstep = rs.indices(R)
            g = self.group
            if step == 1 and rstep == 1 and a % g == 0 and (b % g == 0 or b == K):
                img = self.img[a // g:(b + g - 1) // g, r0:]       # (keeps the rows behind the slice: tiles read 256 rows from r0)
        return X3Weight(self.t[rs, cs], img, self.f16, self.group)


class F8Weight:
    """A bf16 [N][K] FORWARD GEMM operand together with its per-tensor fp8 scale {s, 1/s} (device, fp32, `gast_f8_scale_multi`):
    GAST_HIP_DTYPE=fp8 runs the GEMMs whose weight segments all carry the same scale with e4m3 operands.  Slices keep the scale."""
    __slots__ = ('t', 'scale')

    def __init__(self, t, scale):
        self.t, self.scale = t, scale

    def __getitem__(self, idx):
        return F8Weight(self.t[idx], self.scale)


class Packer:
    def __init__(self, model, spec, named=None):
        """named: explicit (name, tensor) list in `named_parameters()` order -- for nn.DataParallel replicas, whose parameters are
        plain tensor attributes (model/gast_net.py: replica_parameters)."""
        self.spec = spec
        if named is None:
            named = list(model.named_parameters())
        self.params = [p for _, p in named]
        self.names = [n for n, _ in named]
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.goff = []
        off = 0
        for p in self.params:
            self.goff.append(off)
            off += p.numel()
        self.gsize = off
        self.W = Layout()      # act-dtype operands
        self.F = Layout()      # fp32 packed values (bias1, C_k)
        self.S = Layout()      # fp32 packed gradients (same names as W/F regions that have gradients)
        self.copy_jobs, self.fold_jobs, self.unpack_jobs, self.unfold_jobs = [], [], [], []
        self.direct = {}       # engine key -> parameter (read directly / gradient written directly into the grad buffer)
        self._build(model)
        self._dev = {}

    # ------------------------------------------------------------------------------------------ description
    def _pref(self, p, off, rs, cs):
        return Ref(BASE_ABS, p, off, rs, cs)

    def _gref(self, p, off, rs, cs):
        return Ref(BASE_G, None, self.goff[self.index[id(p)]] + off, rs, cs)

    def _block(self, p, poff, R, S, prs, pcs, region, row0, regionT=None, fp32=False, grad=True):
        """logical block B[r][s] = p.flat[poff + r*prs + s*pcs]  ->  region[row0 + r][s]  (and regionT[s][row0 + r])."""
        base = BASE_F if fp32 else BASE_W
        lay = self.F if fp32 else self.W
        self.copy_jobs.append((self._pref(p, poff, prs, pcs), lay.ref(base, region, row0), R, S, fp32))
        if regionT is not None:
            self.copy_jobs.append((self._pref(p, poff, prs, pcs), lay.ref(base, regionT, 0, row0, transposed=True), R, S, fp32))
        if grad:
            self.unpack_jobs.append((self.S.ref(BASE_S, region, row0), self._gref(p, poff, prs, pcs), R, S))

    def _build(self, m):
        sp = self.spec
        J = sp.J
        C0 = sp.channels
        L = len(sp.fw)
        d = self.direct
        d['init_bn.weight'], d['init_bn.bias'] = m.init_bn.weight, m.init_bn.bias
        d['expand_bn.weight'], d['expand_bn.bias'] = m.expand_bn.weight, m.expand_bn.bias
        d['expand_w'] = m.expand_conv.weight
        for s, gab in enumerate(m.layers_graph_conv):
            g = 'g%d.' % s
            C = C0 * 2 ** s
            Ci = C // NHEADS
            N1 = 5 * C + 2 * NHEADS
            loc, glb = gab.local_graph_layer, gab.global_graph_layer
            for name, r, c in ((g + 'Bg1', N1, C), (g + 'Bg1T', C, N1), (g + 'Blc', C, 2 * C), (g + 'BlcT', 2 * C, C),
                               (g + 'Bgc', C, C), (g + 'BgcT', C, C), (g + 'Bbc', 2 * C, 3 * C), (g + 'BbcT', 3 * C, 2 * C)):
                self.W.add(name, r, c)
            for name, r, c in ((g + 'Bg1', N1, C), (g + 'Blc', C, 2 * C), (g + 'Bgc', C, C), (g + 'Bbc', 2 * C, 3 * C),
                               (g + 'bias1', 1, N1), (g + 'C_k', NHEADS, J * J)):
                self.S.add(name, r, c)
            self.F.add(g + 'bias1', 1, N1)
            self.F.add(g + 'C_k', NHEADS, J * J)
            # SemCH weights W (2, Cin, Cout): Bg1 rows n <- W[q][k][n]  (local_attention.py:37-38)
            for q, (mod, row0) in enumerate(((loc.gcn_sym, 0), (loc.gcn_sym, C), (loc.gcn_con, 2 * C), (loc.gcn_con, 3 * C))):
                self._block(mod.W, (q % 2) * C * C, C, C, 1, C, g + 'Bg1', row0, g + 'Bg1T')
            for h, att in enumerate(glb.attentions):
                # g conv1d (Ci, C, 1) rows 4C + h*Ci.., bias -> bias1
                self._block(att.g.weight, 0, Ci, C, C, 1, g + 'Bg1', 4 * C + h * Ci, g + 'Bg1T')
                self._block(att.g.bias, 0, 1, Ci, Ci, 1, g + 'bias1', 0, None, fp32=True, grad=False)
                # the bias block lives at columns 4C + h*Ci of the single bias row
                self.copy_jobs[-1] = (self.copy_jobs[-1][0], self.F.ref(BASE_F, g + 'bias1', 0, 4 * C + h * Ci), 1, Ci, True)
                self.unpack_jobs.append((self.S.ref(BASE_S, g + 'bias1', 0, 4 * C + h * Ci), self._gref(att.g.bias, 0, Ci, 1), 1, Ci))
                self._block(att.C_k, 0, 1, J * J, J * J, 1, g + 'C_k', h, None, fp32=True)
                w = att.concat_project[0].weight          # (1, 2Ci, 1, 1)
                for t, (conv, woff, row) in enumerate(((att.theta, 0, 5 * C + h), (att.phi, Ci, 5 * C + NHEADS + h))):
                    self.fold_jobs.append(dict(W=conv.weight, w=w, woff=woff, b=conv.bias, Ci=Ci, C=C,
                                               row=self.W.ref(BASE_W, g + 'Bg1', row), col=self.W.ref(BASE_W, g + 'Bg1T', 0, row, transposed=True),
                                               bias=self.F.ref(BASE_F, g + 'bias1', 0, row)))
                    self.unfold_jobs.append(dict(dv=self.S.ref(BASE_S, g + 'Bg1', row), da=self.S.ref(BASE_S, g + 'bias1', 0, row),
                                                 W=conv.weight, w=w, woff=woff, b=conv.bias, Ci=Ci, C=C))
            self._block(loc.cat_conv.weight, 0, C, 2 * C, 2 * C, 1, g + 'Blc', 0, g + 'BlcT')
            self._block(glb.cat_conv.weight, 0, C, C, C, 1, g + 'Bgc', 0, g + 'BgcT')
            self._block(gab.cat_conv.weight, 0, 2 * C, 3 * C, 3 * C, 1, g + 'Bbc', 0, g + 'BbcT')
            d[g + 'e_sym'], d[g + 'e_con'] = loc.gcn_sym.e, loc.gcn_con.e
            for key, bn in ((g + 'bn_1', loc.bn_1), (g + 'bn_2', loc.bn_2), (g + 'lcat_bn', loc.cat_bn), (g + 'gcat_bn', glb.cat_bn),
                            (g + 'cat_bn', gab.cat_bn)):
                d[key + '.weight'], d[key + '.bias'] = bn.weight, bn.bias
        for i in range(1, L):
            lk = 'l%d.' % i
            C = C0 * 2 ** i
            k = sp.kw[i]
            conv, conv1 = m.layers_conv[2 * i - 2], m.layers_conv[2 * i - 1]
            self.W.add(lk + 'conv', C, k * C)
            self.W.add(lk + 'convT', k * C, C)
            self.W.add(lk + 'conv1', C, C)
            self.W.add(lk + 'conv1T', C, C)
            self.S.add(lk + 'conv', C, k * C)
            self.S.add(lk + 'conv1', C, C)
            for tap in range(k):   # weight (Cout, Cin, k, 1): conv[n][tap*C + c] = w[n][c][tap];  convT[tap*C + c][n]
                self.copy_jobs.append((self._pref(conv.weight, tap, C * k, k), self.W.ref(BASE_W, lk + 'conv', 0, tap * C), C, C, False))
                self.copy_jobs.append((self._pref(conv.weight, tap, C * k, k), self.W.ref(BASE_W, lk + 'convT', tap * C, 0, transposed=True), C, C, False))
                self.unpack_jobs.append((self.S.ref(BASE_S, lk + 'conv', 0, tap * C), self._gref(conv.weight, tap, C * k, k), C, C))
            self._block(conv1.weight, 0, C, C, C, 1, lk + 'conv1', 0, lk + 'conv1T')
            for key, bn in ((lk + 'bn0', m.layers_bn[2 * i - 2]), (lk + 'bn1', m.layers_bn[2 * i - 1])):
                d[key + '.weight'], d[key + '.bias'] = bn.weight, bn.bias
        CL = C0 * 2 ** L
        self.W.add('shrink', 3, CL)
        self.W.add('shrinkT', CL, 8)       # columns 3..7 stay zero (buffer is zero-initialised)
        self.S.add('shrink', 8, CL)        # the weight-gradient GEMM runs on 8 padded rows
        self._block(m.shrink.weight, 0, 3, CL, CL, 1, 'shrink', 0, 'shrinkT')
        self.direct_index = {k: self.index[id(p)] for k, p in d.items()}

    # ------------------------------------------------------------------------------------------ gradient buckets
    def set_buckets(self, ranges):
        """ranges: [(start, end)] of the flat gradient buffer (gast_hip.dist.FlatGradAllReduce.ranges).  Splits the unpack / unfold
        job lists by the bucket their destination parameter lies in, so that a bucket can be completed (ops.run_unpack(...,
        bucket=i)) and exchanged while the rest of the backward pass is still running."""
        ranges = [tuple(r) for r in ranges]
        if getattr(self, 'bucket_ranges', None) == ranges:
            return
        def bucket_of(off):
            for i, (a, b) in enumerate(ranges):
                if a <= off < b:
                    return i
            raise ValueError('gradient offset %d outside every bucket' % off)
        self.unpack_by_bucket = [[] for _ in ranges]
        self.unfold_by_bucket = [[] for _ in ranges]
        for job in self.unpack_jobs:
            self.unpack_by_bucket[bucket_of(job[1].off)].append(job)
        for j in self.unfold_jobs:
            self.unfold_by_bucket[bucket_of(self.goff[self.index[id(j['W'])]])].append(j)
        self.bucket_ranges = ranges
        for st in self._dev.values():       # device tables of the unbucketed lists stay valid; bucket tables are built on demand
            if st.get('tables'):
                for k in [k for k in st['tables'] if k.startswith('unpack') and ':' in k]:
                    del st['tables'][k]

    # ------------------------------------------------------------------------------------------ buffers
    def f8_regions(self):
        """the forward operands that run in fp8: every packed operand except the transposed twins (input gradients stay bf16) and
        the 3-row output layer"""
        return [n for n in self.W.regions if not n.endswith('T') and n != 'shrink']

    def f8_jobs(self, st):
        return [(self.W.view(st['Wb'], n), st['F8s'][i]) for i, n in enumerate(self.f8_regions())]

    def state(self, dev, dt, x3=False, f8=False, h16=True):
        """Per (device, dtype, x3) persistent buffers + device job tables.  x3 (GAST_F32X3): every packed fp32 operand also gets a
        pre-split bf16 image (`Xb`), refreshed by ops.run_pack after the copy / fold launches."""
        f16fwd = bool(x3) and dt == torch.float32 and x3_forward_f16()
        h16 = bool(h16) and dt != torch.float32 and not f8 and h16_images(dt)
        key = (str(dev), dt, bool(x3), bool(f8), f16fwd, h16)
        st = self._dev.get(key)
        ptrs = tuple(p.data_ptr() for p in self.params)
        if st is None or st['ptrs'] != ptrs:
            st = {'ptrs': ptrs, 'Wb': torch.zeros(self.W.size, dtype=dt, device=dev),
                  'Fb': torch.zeros(self.F.size, dtype=torch.float32, device=dev), 'tables': None, 'Xb': None, 'F8s': None}
            if f8 and dt == torch.bfloat16:
                st['F8s'] = torch.ones(len(self.f8_regions()), 2, dtype=torch.float32, device=dev)
            if x3 and dt == torch.float32:
                X = Layout()
                for n, (_, r, c) in self.W.regions.items():
                    X.add(n, (c + 15) // 16, x3_image_rows(r) * 32)
                st['X'] = X
                st['f16fwd'] = f16fwd
                st['Xb'] = torch.zeros(X.size, dtype=torch.bfloat16, device=dev)
            elif h16:
                # 16-bit modes (round 5): the layout image of every packed operand, 32 values per row -- what the large-M kernel's
                # weight DMA streams (operands whose K is no multiple of 8 get none: their GEMMs stay on the 128 x 128 kernel)
                X = Layout()
                for n, (_, r, c) in self.W.regions.items():
                    X.add(n, (c + 31) // 32, x3_image_rows(r) * 32)
                st['X'] = X
                st['h16img'] = True
                st['Xb'] = torch.zeros(X.size, dtype=dt, device=dev)
            self._dev[key] = st
        return st

    def signature(self):
        """identity of the parameters' current VALUES as far as the host can tell: torch's in-place version counters (optimizer steps,
        load_state_dict, manual edits) + the epoch of raw writers (FlatAdam)"""
        return (PARAM_EPOCH[0],) + tuple(p._version for p in self.params)

    def unchanged(self, st, frozen=False):
        """True if the packed operands in `st` may be reused instead of being rebuilt from the parameters.  Only ever under
        `frozen` -- the caller vouches that this is an inference call (eval mode, no gradient wanted): a training loop repacks on
        every forward, whatever the host believes about the parameters -- and only while the host-visible signature (version
        counters + the epoch of raw writers) is unchanged and no raw writer lives inside a captured graph.  What the signature
        cannot see are in-place writes through `.data` (`p.data.mul_()`, `dist.broadcast(p.data)`): after such a write in an
        inference loop call `model.invalidate_packed()`.  Never inside a stream capture: a replayed graph must refresh the operands
        itself."""
        if not frozen or CAPTURED_WRITER[0] or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            st['packed_sig'] = None
            return False
        sig = self.signature()
        if st.get('packed_sig') == sig:
            return True
        st['packed_sig'] = sig           # (the caller packs now)
        return False

    def invalidate(self):
        """forget what the packed operands were built from: the next forward repacks (after a `.data` write in an inference loop)"""
        for st in self._dev.values():
            st['packed_sig'] = None

    def image_jobs(self, st):
        """(operand view, image view, kind) per packed operand, for ops.run_pack; kind = gast_x3_image_job.f16: 0 bf16 pairs, 1 fp16
        pairs (fp32 operands), 2 the layout image of a 16-bit operand"""
        if st.get('h16img'):
            return [(self.W.view(st['Wb'], n), self._image(st, n), 2) for n in self.W.regions if self._h16_ok(n)]
        return [(self.W.view(st['Wb'], n), self._image(st, n), int(self._f16(st, n))) for n in self.W.regions]

    def _h16_ok(self, n):
        _, r, c = self.W.regions[n]
        return c % 8 == 0 and c >= 8

    @staticmethod
    def _f16(st, n):
        """the forward operands (every packed region but the transposed twins of the input gradients) in fp16 pairs?"""
        return bool(st.get('f16fwd')) and not n.endswith('T')

    def _image(self, st, n):
        g, per = st['X'].regions[n][1:]
        return st['X'].view(st['Xb'], n).view(g, per // 32, 32)

    def inputs(self, st):
        """engine `inp` dict: operand views (act dtype), fp32 packed views, raw parameters."""
        inp = {n: self.W.view(st['Wb'], n) for n in self.W.regions}
        if st.get('h16img'):
            inp = {n: (X3Weight(w, self._image(st, n), False, 32) if self._h16_ok(n) else w) for n, w in inp.items()}
        elif st.get('Xb') is not None:
            inp = {n: X3Weight(w, self._image(st, n), self._f16(st, n)) for n, w in inp.items()}
        if st.get('F8s') is not None:
            for i, n in enumerate(self.f8_regions()):
                inp[n] = F8Weight(inp[n], st['F8s'][i])
        for n in self.F.regions:
            v = self.F.view(st['Fb'], n)
            inp[n] = v.view(-1) if n.endswith('bias1') else v.view(NHEADS, self.spec.J, self.spec.J)
        for k, p in self.direct.items():
            inp[k] = p.detach()
        return inp

    def grad_outputs(self, G, Sb):
        """engine `gout` dict: where every gradient is written (packed scratch views / views into the flat grad buffer)."""
        out = {}
        for n in self.S.regions:
            v = self.S.view(Sb, n)
            if n.endswith('bias1'):
                v = v.view(-1)
            elif n.endswith('C_k'):
                v = v.view(NHEADS, self.spec.J, self.spec.J)
            out[n] = v
        for k, i in self.direct_index.items():
            p = self.params[i]
            out[k] = G[self.goff[i]:self.goff[i] + p.numel()].view(p.shape)
        return out

    def grad_views(self, G):
        return [G[o:o + p.numel()].view(p.shape) for o, p in zip(self.goff, self.params)]
