"""GAST-Net spatio-temporal models on the MI355X HIP plan -- drop-in for the reference's `model/gast_net.py`.

Public surface kept from the reference:
  * `SpatioTemporalModel(adj, num_joints_in, in_features, num_joints_out, filter_widths, causal=False, dropout=0.25,
    channels=64, dense=False)` (reference gast_net.py:113-114) and `SpatioTemporalModelOptimized1f(... same minus dense)`
    (:191-192); `receptive_field()` (:62-69); `forward(x: (B,T,J,in_features)) -> (B,T',J,3)` (:84-104);
  * the `state_dict` key/shape contract (same sub-module names, registration order and initialisers), so the shipped
    checkpoints load with `load_state_dict(strict=True)` and `torch.manual_seed(s)` + construction reproduces the
    reference's initial weights bit for bit;
  * the module-level names `torch`, `nn`, `LocalGraph`, `MultiGlobalGraph`, `SingleGlobalGraph`: the reference's
    `trainval.py:60` reaches `nn` only through `from model.gast_net import *`.

What differs is everything below that surface: forward/backward run as one `torch.autograd.Function` whose body is the
fused launch plan of `gast_hip/engine.py` over hand-written gfx950 kernels.  There is no PyTorch/CPU fallback: calling
the model with CPU tensors, or without `libgast_hip.so`, raises.

Environment knob (never a constructor argument): `GAST_HIP_DTYPE` = `fp32` (default; fp32 MFMA, parity 1e-4), `bf16x3`
(fp32 storage, every GEMM product as bf16 hi/lo split products on the bf16 matrix cores: fp32-class results) or `bf16`
(bf16 activations/weights, fp32 accumulate/statistics).
"""
import collections
import contextlib
import gc
import os
import threading
import weakref

import torch
import torch.nn as nn
from model.local_attention import LocalGraph
from model.global_attention import MultiGlobalGraph, SingleGlobalGraph

from model.local_attention import pattern_table, skeleton_patterns
from gast_hip.engine import Engine, NHEADS


class GraphAttentionBlock(nn.Module):
    """cat(x, LocalGraph(x), MultiGlobalGraph(x)) -> 1x1 conv (3C -> 2C) -> BN -> ReLU (reference gast_net.py:8-33)."""

    def __init__(self, adj, input_dim, output_dim, p_dropout):
        super(GraphAttentionBlock, self).__init__()
        hid_dim = output_dim
        self.relu = nn.ReLU(inplace=True)
        self.local_graph_layer = LocalGraph(adj, input_dim, hid_dim, p_dropout)
        self.global_graph_layer = MultiGlobalGraph(adj, input_dim, input_dim // 4, dropout=p_dropout)
        self.cat_conv = nn.Conv2d(3 * output_dim, 2 * output_dim, 1, bias=False)
        self.cat_bn = nn.BatchNorm2d(2 * output_dim, momentum=0.1)

    def forward(self, x):
        """x: (B, C, T, J) -> (B, 2 C_out, T, J)   (reference :22-33).  Inside SpatioTemporalModel the block is part of the fused
        plan; on its own it runs the same kernels as a forward-only plan (gast_hip/modules.py)."""
        from gast_hip.modules import graph_attention_block_forward
        return graph_attention_block_forward(self, x)


class ModelSpec:
    """Static description of one model instance for the engine."""

    def __init__(self, adj, filter_widths, channels, causal, strided, in_features, dense=False):
        self.J = int(adj.shape[0])
        self.fw = list(filter_widths)
        self.channels = int(channels)
        self.causal = bool(causal)
        self.strided = bool(strided)
        self.dense = bool(dense)
        self.in_features = int(in_features)
        # reference gast_net.py:57,139-143 (dilated) / :215-220 (strided)
        self.pad = [self.fw[0] // 2]
        self.causal_shift = [self.fw[0] // 2 if causal else 0]
        self.dil = [1]
        nd = self.fw[0]
        for i in range(1, len(self.fw)):
            self.pad.append((self.fw[i] - 1) * nd // 2)
            if strided:
                self.causal_shift.append((self.fw[i] // 2) if causal else 0)
            else:
                self.causal_shift.append((self.fw[i] // 2 * nd) if causal else 0)
            self.dil.append(nd)
            nd *= self.fw[i]
        self.receptive_field = 1 + 2 * sum(self.pad)
        # taps of level s's temporal convolution and the frame step between them: fw[s] taps `dil[s]` apart (dilated), fw[s]
        # adjacent taps under a stride of fw[s] (strided), or 2*pad[s]+1 adjacent taps (dense=True ablation, gast_net.py:145-146)
        self.kw = [self.fw[0]] + [2 * self.pad[i] + 1 if self.dense else self.fw[i] for i in range(1, len(self.fw))]
        self.tapstep = [1] + [1 if (self.dense or self.strided) else self.dil[i] for i in range(1, len(self.fw))]
        sym, con = skeleton_patterns(adj)
        self._tab_sym, self.nnz_sym = pattern_table(sym)
        self._tab_con, self.nnz_con = pattern_table(con)
        # max row / column degrees (Dr, Dc) stored right after the CSR/CSC sections of the tables
        off_s, off_c = 2 + 2 * (self.J + 1) + 3 * self.nnz_sym, 2 + 2 * (self.J + 1) + 3 * self.nnz_con
        self.deg_sym = (int(self._tab_sym[off_s]), int(self._tab_sym[off_s + 1]))
        self.deg_con = (int(self._tab_con[off_c]), int(self._tab_con[off_c + 1]))
        self._dev_tabs = {}

    def _tab(self, which, dev):
        key = (which, str(dev))
        if key not in self._dev_tabs:
            src = self._tab_sym if which == 0 else self._tab_con
            self._dev_tabs[key] = src.to(dev)
        return self._dev_tabs[key]

    def pat_sym(self, dev):
        return self._tab(0, dev)

    def pat_con(self, dev):
        return self._tab(1, dev)


def bn_buffers(model):
    """key -> BatchNorm buffers (updated in place by the finalize kernels in train mode)."""
    bufs = {}

    def add(key, bn):
        if not bn.track_running_stats or bn.momentum is None or not bn.affine:
            raise NotImplementedError('gast_net (MI355X build): %s needs affine=True, track_running_stats=True and a numeric momentum '
                                      '(the reference constructs BatchNorm2d(momentum=0.1))' % key)
        # momentum / eps are read from the module on every call, so a VideoPose3D-style `bn.momentum = m` decay schedule works
        bufs[key] = {'running_mean': bn.running_mean, 'running_var': bn.running_var, 'num_batches_tracked': bn.num_batches_tracked,
                     'momentum': float(bn.momentum), 'eps': float(bn.eps)}
    add('init_bn', model.init_bn)
    add('expand_bn', model.expand_bn)
    for s, gab in enumerate(model.layers_graph_conv):
        g = 'g%d.' % s
        loc, glb = gab.local_graph_layer, gab.global_graph_layer
        for key, bn in ((g + 'bn_1', loc.bn_1), (g + 'bn_2', loc.bn_2), (g + 'lcat_bn', loc.cat_bn), (g + 'gcat_bn', glb.cat_bn),
                        (g + 'cat_bn', gab.cat_bn)):
            add(key, bn)
    for i in range(len(model.layers_conv) // 2):
        add('l%d.bn0' % (i + 1), model.layers_bn[2 * i])
        add('l%d.bn1' % (i + 1), model.layers_bn[2 * i + 1])
    return bufs


def replica_parameters(model):
    """(name, tensor) of a DataParallel replica in `named_parameters()` order.  `Module._replicate_for_data_parallel` empties
    `_parameters`; torch.nn.parallel.replicate then sets every broadcast copy as a plain attribute and records it, in registration
    order, in the replica module's `_former_parameters`."""
    out = []
    for mname, mod in model.named_modules():
        former = getattr(mod, '_former_parameters', None)
        if former is None:
            raise RuntimeError('gast_net (MI355X build): %r is marked as a replica but carries no _former_parameters; use '
                               'torch.nn.DataParallel / torch.nn.parallel.replicate, or one process per GPU' % (mname or 'model'))
        for k, t in former.items():
            out.append(((mname + '.' if mname else '') + k, t))
    return out


class _GastFunction(torch.autograd.Function):
    """forward/backward of the whole spatio-temporal path as one autograd node.  Inputs: x and the raw parameters (in
    `model.parameters()` order); the packed operands are produced by one pack launch, the parameter gradients by one unpack
    launch out of a single flat fp32 buffer."""

    @staticmethod
    def forward(ctx, runner, x, training, packer, st, bufs, engine, sink, need_grad, *params):
        if ctx.needs_input_grad[1]:
            raise RuntimeError('gast_net (MI355X build): the gradient with respect to the input batch is not implemented (the '
                               'reference never asks for it); pass x with requires_grad=False')
        ops = engine.ops
        # (opt-in: an evaluation loop over weights the caller declared frozen -- model.freeze_packed() -- packs once)
        if not packer.unchanged(st, frozen=runner.pack_reuse and not training and not need_grad):
            ops.run_pack(packer, st)
        inp = st.get('inp')
        if inp is None:
            inp = st['inp'] = packer.inputs(st)
        engine.centered = runner.centered
        ops.x3 = runner.x3
        ops.f8 = runner.f8
        drop, seed_job = runner.dropout_state(training, x.device)
        # the pass prologue (ONE launch): zero arena, seed bump, and whatever a FlatGradAllReduce.zero_(defer=True) left pending
        pred, sv = engine.forward(x, inp, bufs, training, runner.act_dtype, drop, need_grad=need_grad,
                                  prep={'seed': seed_job, 'zero': runner.take_pending_zero(x.device)})
        ctx.engine, ctx.packer, ctx.st, ctx.inp, ctx.sv, ctx.sink, ctx.runner = engine, packer, st, inp, sv, sink, runner
        ctx.graph_entry = runner.__dict__.pop('_eager_entry', None)      # (the graph-cache entry this eager call warms up, if any)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        engine, packer, st = ctx.engine, ctx.packer, ctx.st
        if ctx.graph_entry is not None:
            ctx.graph_entry.bwd_calls += 1       # an eager backward of this shape has run: its lazy tables / arenas exist
        if ctx.sv is None:
            raise RuntimeError('gast_net (MI355X build): backward through the same forward a second time is not supported (the saved '
                               'activations are released by the first backward, retain_graph has no effect); run forward again')
        dev = dpred.device
        sink = ctx.sink
        if sink is not None:
            ctx.runner.flush_pending_zero(sink)
        # Every gradient kernel ACCUMULATES into its destination (split-M atomics, += for the directly written BatchNorm / e /
        # expand gradients, accumulate-mode unpack): G is either a fresh zero buffer or the caller's flat gradient buffer
        # (FlatGradAllReduce / FlatAdam), which then sums over backward calls like autograd's .grad does.  With such a sink the
        # gradients are NOT returned to autograd (hooks / torch.autograd.grad see None): they are already where p.grad points.
        with torch.cuda.device(dev) if dev.type == 'cuda' else contextlib.nullcontext():
            # (zero-filled by the backward's pass prologue, together with its arena: engine.backward(prep=))
            # f16 mode: every gradient of this pass comes out multiplied by the loss scale (engine.backward) -- into a private buffer,
            # unscaled as it is added to its destination
            scale = engine.loss_scale(ctx.sv['dt'])
            priv = sink is None or scale != 1.0
            G = torch.empty(packer.gsize, dtype=torch.float32, device=dev) if priv else sink
            Sb = torch.empty(packer.S.size, dtype=torch.float32, device=dev)
            prep = {'zero': [Sb] + ([G] if priv else [])}
            gout = packer.grad_outputs(G, Sb)
            gs = ctx.runner.grad_sync if sink is not None else None
            if gs is not None and len(gs.ranges) > 1 and gs.flat is sink and scale != 1.0:
                raise NotImplementedError('gast_net (MI355X build): the bucketed gradient exchange is not available in GAST_HIP_DTYPE=f16 '
                                          '(loss-scaled gradients are unscaled in one pass at the end of backward); use buckets=1')
            if gs is not None and len(gs.ranges) > 1 and gs.flat is sink:
                # bucketed exchange (gast_hip/dist.py): complete and hand over each bucket of the flat buffer as soon as its stage is done
                packer.set_buckets(gs.ranges)

                def stage_done(s_):
                    b = gs.bucket_of_stage(s_)
                    engine.ops.run_unpack(packer, st, Sb, G, True, bucket=b)
                    gs.bucket_ready(b)
                engine.backward(ctx.sv, ctx.inp, dpred.contiguous(), gout, stage_done=stage_done, prep=prep)
                ctx.sv = None
                last = len(gs.ranges) - 1
                engine.ops.run_unpack(packer, st, Sb, G, True, bucket=last)
                gs.bucket_ready(last)
            else:
                engine.backward(ctx.sv, ctx.inp, dpred.contiguous(), gout, prep=prep)
                ctx.sv = None
                engine.ops.run_unpack(packer, st, Sb, G, True)
                if scale != 1.0:
                    if sink is not None:
                        sink.add_(G, alpha=1.0 / scale)
                    else:
                        G.mul_(1.0 / scale)
        if sink is not None:
            return (None,) * 9 + (None,) * len(packer.params)
        return (None,) * 9 + tuple(packer.grad_views(G))


def _capture_kwargs():
    """keyword arguments of torch.cuda.graph() for the module's own captures: with a process group alive, RCCL's watchdog thread polls
    its work events at any time, which the default GLOBAL capture mode turns into an error in THAT thread (and torch then terminates
    the process); thread-local mode only polices the capturing thread"""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return {'capture_error_mode': 'thread_local'}
    return {}


class _GraphEntry:
    """GAST_HIP_GRAPH=1: the captured forward (and backward) hipGraphs of one (shape, mode, arithmetic) of a model, their static
    input / output buffers and the activations the backward graph reads (all in the graphs' private memory pool)."""

    WARMUP = 2      # eager calls before the capture (lazy workspaces, job tables, zero arenas are created by them)

    def __init__(self):
        self.calls = 0
        self.fwd = self.bwd = None
        self.gen = 0
        self.x = self.pred = self.dpred = self.G = self.sink = None
        self.packer = self.runner = None
        self.frozen = False
        self.buf_ptrs = None        # addresses of the BatchNorm buffers the captured kernels update
        self.st = self.ops = None   # the packed-operand state (refreshed eagerly when the parameters changed) and the op set
        self.keep = None
        self.token_ref = None       # weak reference to the token of the forward whose backward has not run yet
        self.bwd_calls = 0          # eager backward passes seen for this key (the capture of a backward graph needs one: lazy state)

    def busy(self):
        """a replayed forward is still waiting for its backward (its autograd node is alive and unconsumed)"""
        return self.token_ref is not None and self.token_ref() is not None


class _Token:
    """lives exactly as long as the autograd node of one replayed forward"""
    __slots__ = ('__weakref__',)


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a stream is capturing: a collection that happens to run inside the capture may finalize
    objects of EARLIER captures (graphs, their private pools) whose destructors free device memory -- prohibited during a capture,
    and an error inside a destructor terminates the process (seen once in the GPU suite: 'Fatal Python error: Aborted' with the
    collector on the stack, in the middle of engine.backward under capture).  Reference counting is unaffected."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _capture_graphs(entry, runner, x, training, packer, st, bufs, engine, sink, need_grad):
    """Capture what _GastFunction.forward / .backward launch for this call as two hipGraphs sharing one memory pool: the saved
    activations of the forward capture are the operands of the backward capture.  Nothing runs here; the caller replays."""
    with _no_gc():
        _capture_graphs_body(entry, runner, x, training, packer, st, bufs, engine, sink, need_grad)


def _capture_graphs_body(entry, runner, x, training, packer, st, bufs, engine, sink, need_grad):
    dev = x.device
    ops = engine.ops
    entry.x = x.clone()
    entry.packer, entry.sink, entry.runner = packer, sink, runner
    entry.frozen = runner.pack_reuse and not training and not need_grad
    pool = torch.cuda.graph_pool_handle()
    engine.centered = runner.centered
    ops.x3 = runner.x3
    ops.f8 = runner.f8
    # the split-K workspace is keyed by launch stream (gast_hip/binding.py): create the capture stream's one HERE, outside the capture,
    # so that it lives in the ordinary allocator and not in (and pinned by) the first captured entry's private pool
    if torch.cuda.graph.default_capture_stream is None:
        torch.cuda.graph.default_capture_stream = torch.cuda.Stream()
    with torch.cuda.stream(torch.cuda.graph.default_capture_stream):
        ops._splitk_ws(dev)
    # parameter packing stays OUTSIDE the captured graphs: _GraphedFunction.forward launches its three kernels eagerly, and only when
    # the parameters changed since the operands were last packed (a training loop: every step; an evaluation loop: once)
    entry.st, entry.ops = st, ops
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g, pool=pool, **_capture_kwargs()):
        inp = st.get('inp')
        if inp is None:
            inp = st['inp'] = packer.inputs(st)
        drop, seed_job = runner.dropout_state(training, dev)
        pred, sv = engine.forward(entry.x, inp, bufs, training, runner.act_dtype, drop, need_grad=need_grad, prep={'seed': seed_job})
    entry.fwd, entry.pred = g, pred
    entry.keep = (sv, inp, pool)
    if need_grad:
        entry.dpred = torch.zeros_like(pred)
        gb = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(gb, pool=pool, **_capture_kwargs()):
            scale = engine.loss_scale(sv['dt'])
            priv = sink is None or scale != 1.0
            G = torch.empty(packer.gsize, dtype=torch.float32, device=dev) if priv else sink
            Sb = torch.empty(packer.S.size, dtype=torch.float32, device=dev)
            engine.backward(sv, inp, entry.dpred, packer.grad_outputs(G, Sb), prep={'zero': [Sb] + ([G] if priv else [])})
            ops.run_unpack(packer, st, Sb, G, True)
            if scale != 1.0:          # (f16 mode: loss-scaled gradients, see _GastFunction.backward)
                if sink is not None:
                    sink.add_(G, alpha=1.0 / scale)
                else:
                    G.mul_(1.0 / scale)
        entry.bwd, entry.G = gb, G
        entry.keep += (Sb,)


class _GraphedFunction(torch.autograd.Function):
    """The same autograd node as _GastFunction with its kernels replayed from the captured hipGraphs (GAST_HIP_GRAPH=1)."""

    @staticmethod
    def forward(ctx, entry, x, *params):
        if ctx.needs_input_grad[1]:
            raise RuntimeError('gast_net (MI355X build): the gradient with respect to the input batch is not implemented (the '
                               'reference never asks for it); pass x with requires_grad=False')
        entry.x.copy_(x)
        for t in entry.runner.take_pending_zero():     # (a deferred FlatGradAllReduce.zero_(): not part of the captured forward)
            t.zero_()
        # (a training / gradient-enabled entry repacks on every call; an inference entry when the parameters changed: Packer.unchanged)
        if not entry.packer.unchanged(entry.st, frozen=entry.frozen):
            entry.ops.run_pack(entry.packer, entry.st)
        entry.fwd.replay()
        entry.gen += 1
        ctx.entry, ctx.gen = entry, entry.gen
        if entry.bwd is not None:
            # the captured graphs keep ONE set of activations: until this call's backward has run (or its autograd node has been
            # freed) further calls of the same shape take the eager path (SpatioTemporalModelBase.forward checks `busy()`)
            ctx.token = _Token()
            entry.token_ref = weakref.ref(ctx.token)
        return entry.pred.clone()

    @staticmethod
    def backward(ctx, dpred):
        e = ctx.entry
        if e.bwd is None:
            raise RuntimeError('gast_net (MI355X build, GAST_HIP_GRAPH=1): this forward was captured without gradients')
        if ctx.gen != e.gen:
            raise RuntimeError('gast_net (MI355X build, GAST_HIP_GRAPH=1): another forward of the same shape ran since the one this '
                               'backward belongs to; the captured graphs keep ONE set of activations (call backward before the next '
                               'forward, or unset GAST_HIP_GRAPH)')
        e.dpred.copy_(dpred)
        if e.sink is not None:
            e.runner.flush_pending_zero(e.sink)
        e.bwd.replay()
        e.token_ref = None          # the activations may be overwritten by the next forward of this shape
        if e.sink is not None:
            return (None, None) + (None,) * len(e.packer.params)
        return (None, None) + tuple(e.packer.grad_views(e.G.clone()))      # (a copy: the next replay rewrites G in place)


class _Runner:
    """Per-model glue: op set, activation dtype, dropout stream."""

    def __init__(self, spec, p_dropout):
        self.spec = spec
        self.p_dropout = float(p_dropout)
        self._engine = None
        self._engines = {}        # nn.DataParallel replicas: one engine (zero arena, queues, eval tables) per device
        self._lock = threading.Lock()
        self._packer = None
        self.grad_sink = None     # optional flat fp32 buffer (model.parameters() order) that backward accumulates into directly
        self.grad_sync = None     # optional gast_hip.dist.FlatGradAllReduce in bucketed mode: told when a bucket of grad_sink is complete
        self.pending_zero = []    # buffers whose zero fill rides in the next forward's pass prologue (FlatGradAllReduce.zero_(defer=True))
        self._seeds = {}
        self._ops_factory = None  # see set_ops(): None in the product
        # Reuse of the packed GEMM operands across inference calls is OPT-IN (model.freeze_packed() / GAST_PACK_REUSE=1):
        # the host cannot see every parameter write (`.data`, a torch optimizer replayed inside the user's own CUDA graph, an EMA swap), and
        # a stale operand silently corrupts validation numbers, while repacking costs 30-95 us per forward.
        self.pack_reuse = os.environ.get('GAST_PACK_REUSE', '0') not in ('0', '')
        # Forward / backward of every (shape, mode, arithmetic, BatchNorm momentum, dropout p) are replayed from hipGraphs captured on
        # the third call, so an unchanged training loop (model(x); loss.backward()) runs at the replay speed instead of paying ~140
        # Python launches (3.9 vs 4.9 ms per step at B = 128).  ON by default since round 3 (GAST_HIP_GRAPH=0 turns it off): the
        # cases a replay cannot serve fall back to the eager path by themselves -- a call inside somebody else's stream capture, a
        # second forward of a shape whose previous forward still awaits its backward, DataParallel replicas, the bucketed exchange
        self.graph_mode = os.environ.get('GAST_HIP_GRAPH', '1') not in ('0', '')
        # least-recently-used cache of (shape, mode, arithmetic) -> _GraphEntry, bounded: a captured entry pins its private memory
        # pool (every saved activation of that shape), and the reference's evaluate() feeds whole videos of many different lengths
        # (main.py:299-353) -- an unbounded cache would grow by one pool per video length.  The evicted entry's graphs and pool are
        # released with it; a shape that keeps being evicted before its third call simply stays on the eager path.
        self._graphs = collections.OrderedDict()
        self.graph_cache_max = max(1, int(os.environ.get('GAST_HIP_GRAPH_MAX', '8')))

    def __getstate__(self):
        return {'spec': self.spec, 'p_dropout': self.p_dropout, '_engine': None, '_engines': {}, '_packer': None, 'grad_sink': None,
                'grad_sync': None, 'pending_zero': [], '_seeds': {}, '_ops_factory': self._ops_factory, 'pack_reuse': self.pack_reuse, 'graph_mode': self.graph_mode,
                '_graphs': collections.OrderedDict(), 'graph_cache_max': self.graph_cache_max}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._lock = threading.Lock()

    def set_ops(self, factory):
        """TEST SEAM, explicit.  The product has ONE op set, gast_hip.binding.HipOps (device tensors, HIP kernels, no fallback), and
        nothing in the product calls this method.  The test-suite checks the host plan on CPU by handing a model's runner a numpy
        mirror of that op set (tests/fake_backend.py: use_oracle_ops(model)); `bench.py --dry-run-cpu` does the same for the launcher
        dry run.  No import look-up, no environment switch: a runner nobody called set_ops() on cannot leave the HIP path."""
        self._ops_factory = factory
        self._engine = None
        self._engines = {}

    @property
    def ops_factory(self):
        return self._ops_factory

    @property
    def act_dtype(self):
        v = os.environ.get('GAST_HIP_DTYPE', 'fp32').lower()
        if v in ('fp32', 'f32', 'float32', 'bf16x3', 'x3'):
            return torch.float32
        if v in ('bf16', 'bfloat16', 'fp8'):
            return torch.bfloat16
        if v in ('f16', 'fp16', 'float16', 'half'):
            # IEEE binary16 storage and matrix operands (libgast_hip_f16.so; fp32 accumulate, statistics, softmax, master weights and
            # parameter gradients as in bf16 mode): 11 significand bits instead of 8 -- the 16-bit mode that meets the north star's
            # 1e-2 bound (bf16 cannot: tests/test_bf16_floor_cpu.py); gradients travel multiplied by a power-of-two loss scale
            return torch.float16
        raise ValueError('GAST_HIP_DTYPE must be fp32, bf16x3, bf16, f16 or fp8, got %r' % v)

    @property
    def x3(self):
        """GAST_HIP_DTYPE=bf16x3: fp32 storage, GEMMs and weight gradients on split-bf16 MFMA products (include/gast_hip.h)."""
        return os.environ.get('GAST_HIP_DTYPE', 'fp32').lower() in ('bf16x3', 'x3')

    @property
    def f8(self):
        """GAST_HIP_DTYPE=fp8 (BASELINE.json configs[4], "mixed fp8 channel GEMMs"): bf16 storage; the FORWARD channel GEMMs run with
        OCP e4m3 operands (per-tensor power-of-two weight scales) on v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulation; input and weight
        gradients, statistics, softmax and master weights as in bf16 mode."""
        return os.environ.get('GAST_HIP_DTYPE', 'fp32').lower() == 'fp8'

    @property
    def centered(self):
        """Store pre-BN tensors as x - running_mean (GAST_HIP_CENTER=1; default off).  Removes the |mean|/std factor from the
        bf16 rounding of a stored channel; measured gain at the BASELINE size is ~8 % of the RMS drift,
        and a running_mean that does not match the data (fresh fine-tuning set) would make it worse, hence opt-in."""
        return os.environ.get('GAST_HIP_CENTER', '0').lower() not in ('0', 'off', 'false', '')

    def _new_engine(self):
        if self.ops_factory is not None:
            ops = self.ops_factory()
        else:
            from gast_hip.binding import HipOps
            ops = HipOps()
        return Engine(self.spec, ops)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = self._new_engine()
        return self._engine

    def engine_for(self, dev):
        """The engine of a DataParallel replica on `dev`: replicas run concurrently in threads and share this runner through the
        shallow __dict__ copy of `_replicate_for_data_parallel`, so each device gets its own engine and op set (the mutable state of
        a pass lives there)."""
        key = str(dev)
        with self._lock:
            eng = self._engines.get(key)
            if eng is None:
                eng = self._engines[key] = self._new_engine()
        return eng

    def take_pending_zero(self, dev=None):
        """the buffers whose deferred zero fill rides in this forward's pass prologue; a buffer on ANOTHER device than the pass (dev) is
        zeroed right away instead (the prologue launch runs on the pass's device and stream)"""
        z, self.pending_zero = self.pending_zero, []
        if dev is not None:
            for t in z:
                if t.device != dev:
                    t.zero_()
            z = [t for t in z if t.device == dev]
        return z

    def flush_pending_zero(self, t):
        """A backward pass is about to ACCUMULATE into `t`: a zero fill that zero_(defer=True) parked for "the next forward" and that no
        forward has consumed (a retained graph, an exception between zero_() and the forward, a sink shared with another model) happens
        now -- otherwise the buffer would not be zeroed before the accumulation and a LATER forward would wipe the accumulated sum."""
        if any(p is t for p in self.pending_zero):
            self.pending_zero = [p for p in self.pending_zero if p is not t]
            t.zero_()

    def dropout_state(self, training, dev):
        """A fresh dropout stream per training forward: the seed lives on the device (graph-capture friendly).  Returns (Dropout
        whose seed is this pass's copy -- NOT yet written --, (counter, copy)): the bump `*copy = ++*counter` is part of the
        forward's pass prologue (engine.forward(prep=): one launch with the arena's zero fill instead of an add + a clone)."""
        if not training or self.p_dropout <= 0:
            return None, None
        from gast_hip.binding import Dropout, dropout_params
        thresh, inv_keep = dropout_params(self.p_dropout)
        key = str(dev)
        with self._lock:
            if key not in self._seeds:
                s = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
                self._seeds[key] = torch.tensor([s], dtype=torch.int32, device=dev)
        seed = self._seeds[key]
        copy = torch.empty_like(seed)
        return Dropout(copy, thresh, inv_keep), (seed, copy)


class SpatioTemporalModelBase(nn.Module):
    """
    Do not instantiate this class.
    """

    def __init__(self, adj, num_joints_in, in_features, num_joints_out,
                 filter_widths, causal, dropout, channels):
        super().__init__()
        for fw in filter_widths:
            assert fw % 2 != 0, 'Only odd filter widths are supported'
        self.num_joints_in = num_joints_in
        self.in_features = in_features
        self.num_joints_out = num_joints_out
        self.filter_widths = filter_widths
        self.drop = nn.Dropout(dropout)
        self.relu = nn.ReLU(inplace=True)
        self.pad = [filter_widths[0] // 2]
        self.init_bn = nn.BatchNorm2d(in_features, momentum=0.1)
        self.expand_bn = nn.BatchNorm2d(channels, momentum=0.1)
        self.shrink = nn.Conv2d(2 ** len(self.filter_widths) * channels, 3, 1, bias=False)

    def _finish_init(self, adj, filter_widths, causal, dropout, channels, strided, dense=False):
        """Common tail of both variants' constructors: temporal stack + engine spec (reference :133-157 / :210-233)."""
        layers_conv, layers_graph_conv, layers_bn = [], [], []
        layers_graph_conv.append(GraphAttentionBlock(adj, channels, channels, p_dropout=dropout))
        self.causal_shift = [(filter_widths[0]) // 2 if causal else 0]
        next_dilation = filter_widths[0]
        for i in range(1, len(filter_widths)):
            width = 2 ** i * channels
            self.pad.append((filter_widths[i] - 1) * next_dilation // 2)
            if strided:
                self.causal_shift.append((filter_widths[i] // 2) if causal else 0)
                conv = nn.Conv2d(width, width, (filter_widths[i], 1), stride=(filter_widths[i], 1), bias=False)
            else:
                self.causal_shift.append((filter_widths[i] // 2 * next_dilation) if causal else 0)
                conv = nn.Conv2d(width, width, (filter_widths[i], 1) if not dense else (2 * self.pad[-1] + 1, 1),
                                 dilation=(next_dilation, 1) if not dense else (1, 1), bias=False)
            layers_conv.append(conv)
            layers_bn.append(nn.BatchNorm2d(width, momentum=0.1))
            layers_conv.append(nn.Conv2d(width, width, 1, dilation=1, bias=False))
            layers_bn.append(nn.BatchNorm2d(width, momentum=0.1))
            layers_graph_conv.append(GraphAttentionBlock(adj, width, width, p_dropout=dropout))
            next_dilation *= filter_widths[i]
        self.layers_conv = nn.ModuleList(layers_conv)
        self.layers_bn = nn.ModuleList(layers_bn)
        self.layers_graph_conv = nn.ModuleList(layers_graph_conv)
        if channels % 4 != 0:
            raise ValueError('channels must be a multiple of 4 (4 attention heads, reference gast_net.py:16)')
        spec = ModelSpec(adj, filter_widths, channels, causal, strided, self.in_features, dense=dense)
        # kept out of nn.Module's registries (no parameters / buffers of its own -> state_dict is untouched)
        object.__setattr__(self, '_runner', _Runner(spec, dropout))

    def freeze_packed(self, on=True):
        """Opt in (or out) to reusing the GEMM-ready operands across INFERENCE calls (eval mode under torch.no_grad()): they are then
        rebuilt only when the host sees the parameters change (optimizer steps, load_state_dict, tracked in-place edits, this library's
        FlatAdam).  Off by default: every forward repacks (30-95 us), which is always correct.  What the host cannot see once you opted
        in -- writes through `.data`, a torch optimizer replayed inside your own CUDA graph, `dist.broadcast(p.data)`, an EMA swap --
        needs `invalidate_packed()` after the write.  Training-mode and gradient-enabled forwards always repack."""
        self._runner.pack_reuse = bool(on)
        if not on:
            self.invalidate_packed()
        return self

    def invalidate_packed(self):
        """Force the next forward to rebuild the GEMM-ready operands from the parameters.  Only needed after `freeze_packed()`, when the
        parameters were written in a way that leaves no trace the host could see (see there)."""
        if self._runner._packer is not None:
            self._runner._packer.invalidate()

    def receptive_field(self):
        """
        Return the total receptive field of this model as # of frames.
        """
        frames = 0
        for f in self.pad:
            frames += f
        return 1 + 2 * frames

    def total_causal_shift(self):
        """
        Return the asymmetric offset for sequence padding (kept for API parity, reference gast_net.py:71-82).
        """
        frames = self.causal_shift[0]
        next_dilation = self.filter_widths[0]
        for i in range(1, len(self.filter_widths)):
            frames += self.causal_shift[i] * next_dilation
            next_dilation *= self.filter_widths[i]
        return frames

    def forward(self, x):
        """
        X: (B, T, N, C)  -- batch, frames, keypoints, features per keypoint (the reference's docstring says (B,C,T,N),
        its asserts and callers use this layout: gast_net.py:93-95, main.py:219-230).
        """
        assert len(x.shape) == 4
        assert x.shape[-2] == self.num_joints_in
        assert x.shape[-1] == self.in_features
        runner = self._runner
        if not x.is_cuda and runner.ops_factory is None:
            raise RuntimeError('gast_net (MI355X build): input is on %s. This implementation has no CPU fallback; move the '
                               'model and the batch to the GPU (`.cuda()`).' % x.device)
        x = x.contiguous().float()
        from gast_hip.packer import Packer
        with torch.cuda.device(x.device) if x.is_cuda else contextlib.nullcontext():    # kernels go to the stream of x's device
            if getattr(self, '_is_replica', False):
                # nn.DataParallel replica (reference trainval.py:56-61): its parameters are the broadcast, non-leaf copies held as
                # plain attributes (`_parameters` is empty), fresh on every forward, and the runner object is shared with the other
                # replicas' threads -- so: parameters by attribute walk, packing tables rebuilt for this call, a per-device engine,
                # gradients returned to autograd (which reduces them onto the source device).  This is the compatibility path;
                # one process per GPU with gast_hip.dist.FlatGradAllReduce is the fast one.
                named = replica_parameters(self)
                packer = Packer(self, runner.spec, named=named)
                engine, sink = runner.engine_for(x.device), None
            else:
                first = next(self.parameters())
                if first.device != x.device:
                    raise RuntimeError('gast_net (MI355X build): the model is on %s, the batch on %s' % (first.device, x.device))
                if runner._packer is None or runner._packer.params[0] is not first:
                    runner._packer = Packer(self, runner.spec)
                    runner._graphs = collections.OrderedDict()          # (captured graphs hold the old parameters' addresses)
                packer = runner._packer
                engine, sink = runner.engine, runner.grad_sink
            if runner.act_dtype != torch.float32 and hasattr(engine.ops, 'set_h16'):
                engine.ops.set_h16(runner.act_dtype)        # (which flavour of the library the launches of this call go to)
            st = packer.state(x.device, runner.act_dtype, x3=runner.x3 and runner.ops_factory is None,
                              f8=runner.f8 and runner.ops_factory is None, h16=runner.ops_factory is None)
            # (inside an autograd.Function grad mode is off and needs_input_grad ignores torch.no_grad(): decided here)
            need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in packer.params)
            gs = runner.grad_sync
            if (runner.graph_mode and x.is_cuda and runner.ops_factory is None and not getattr(self, '_is_replica', False)
                    and not (gs is not None and len(gs.ranges) > 1)       # (the bucketed exchange hooks into the eager backward)
                    and not torch.cuda.is_current_stream_capturing()):    # (inside a caller's capture the kernels join THAT graph)
                bufs_now = bn_buffers(self)
                key = (tuple(x.shape), self.training, need_grad, str(x.device), runner.act_dtype, runner.x3, bool(st.get('f16fwd')), runner.f8,
                       runner.centered,
                       None if sink is None else sink.data_ptr(), runner.p_dropout if self.training else 0.0,
                       tuple((b['momentum'], b['eps']) for b in bufs_now.values()))      # (baked into a capture: part of its identity)
                entry = runner._graphs.get(key)
                buf_ptrs = tuple(t.data_ptr() for b in bufs_now.values() for t in (b['running_mean'], b['running_var'], b['num_batches_tracked']))
                if entry is not None and entry.fwd is not None and (entry.st is not st or entry.buf_ptrs != buf_ptrs):
                    # the parameters moved to other memory since the capture (FlatAdam re-homes them into its flat buffer, .data was
                    # re-bound, ...) or a BatchNorm buffer was re-bound (the captured finalize kernels update running statistics at the
                    # captured addresses -- ADVICE round 3): the captured graphs are stale -- drop them and warm up again
                    del runner._graphs[key]
                    entry = None
                if entry is None:
                    # a key that differs from a cached one ONLY in what is baked into the capture besides shape and mode (BatchNorm
                    # momentum / eps, dropout p, the gradient sink) supersedes it: a momentum schedule would otherwise strand one
                    # activation pool per value until the LRU bound is reached
                    for k_old in [k for k in runner._graphs if k[:9] == key[:9]]:
                        del runner._graphs[k_old]
                    entry = runner._graphs[key] = _GraphEntry()
                    entry.buf_ptrs = buf_ptrs
                    while len(runner._graphs) > runner.graph_cache_max:
                        runner._graphs.popitem(last=False)       # least recently used: its graphs, static buffers and pool go with it
                else:
                    runner._graphs.move_to_end(key)
                entry.calls += 1
                # capture after WARMUP eager forwards AND -- when a backward graph is wanted -- at least one eager backward of this
                # key (its job tables / arenas are created lazily: a loop that only ever calls forward with autograd enabled stays eager)
                if entry.calls > _GraphEntry.WARMUP and not entry.busy() and (not need_grad or entry.fwd is not None or entry.bwd_calls > 0):
                    if entry.fwd is None:
                        _capture_graphs(entry, runner, x, self.training, packer, st, bufs_now, engine, sink, need_grad)
                    return _GraphedFunction.apply(entry, x, *packer.params)
                if entry.fwd is None:
                    runner._eager_entry = entry
            return _GastFunction.apply(runner, x, self.training, packer, st, bn_buffers(self), engine, sink, need_grad, *packer.params)


class SpatioTemporalModel(SpatioTemporalModelBase):
    """
    Reference 3D pose estimation model with temporal (dilated) convolutions; usable for all use-cases.
    """

    def __init__(self, adj, num_joints_in, in_features, num_joints_out,
                 filter_widths, causal=False, dropout=0.25, channels=64, dense=False):
        """
        Arguments (identical to the reference, gast_net.py:113-128):
        num_joints_in -- number of input joints (e.g. 17 for Human3.6M)
        in_features -- number of input features for each joint (typically 2 for 2D input)
        num_joints_out -- number of output joints (can be different than input)
        filter_widths -- list of convolution widths, which also determines the # of blocks and receptive field
        causal -- use causal convolutions instead of symmetric convolutions (for real-time applications)
        dropout -- dropout probability
        channels -- number of convolution channels
        dense -- use regular dense convolutions instead of dilated convolutions (ablation experiment)
        """
        super().__init__(adj, num_joints_in, in_features, num_joints_out, filter_widths, causal, dropout, channels)
        self.expand_conv = nn.Conv2d(in_features, channels, (filter_widths[0], 1), bias=False)
        nn.init.kaiming_normal_(self.expand_conv.weight)
        self._finish_init(adj, filter_widths, causal, dropout, channels, strided=False, dense=dense)


class SpatioTemporalModelOptimized1f(SpatioTemporalModelBase):
    """
    Single-frame-batching variant (input length = receptive field, output length = 1): strided instead of dilated
    convolutions, weights interchangeable with SpatioTemporalModel (reference gast_net.py:180-251).
    """

    def __init__(self, adj, num_joints_in, in_features, num_joints_out,
                 filter_widths, causal=False, dropout=0.25, channels=64):
        super().__init__(adj, num_joints_in, in_features, num_joints_out, filter_widths, causal, dropout, channels)
        self.expand_conv = nn.Conv2d(in_features, channels, (filter_widths[0], 1), stride=(filter_widths[0], 1), bias=False)
        nn.init.kaiming_normal_(self.expand_conv.weight)
        self._finish_init(adj, filter_widths, causal, dropout, channels, strided=True)
