#!/usr/bin/env python3
"""Benchmark of the GAST-Net spatio-temporal hot path on MI355X (BASELINE.json metric: sequences/sec, fwd+bwd).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = the reference's training step on one synthetic batch (reference main.py:227-239): zero_grad -> forward ->
mpjpe -> backward -> [gradient all-reduce over RCCL when N>1] -> Adam(amsgrad) step, on
BASELINE.json configs[1]: SpatioTemporalModel, J=17, filter_widths 3,3,3 (27-frame receptive field), channels=128,
B=128 sequences of T=27 frames per GPU (weak scaling), dropout 0.05.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  `--gpus N` without a torchrun environment re-launches itself through torch.distributed.run
(one rank per GPU, 127.0.0.1 rendezvous).

Arithmetic (`--dtype`, default bf16x3): every GEMM / weight-gradient product runs on the bf16 matrix cores
(v_mfma_f32_32x32x16_bf16, fp32 accumulation).  `bf16x3` keeps storage in fp32 and forms each product from bf16 hi/lo pairs
(hi*hi + hi*lo + lo*hi): it is the mode that meets the north-star tolerance in TRAIN mode at this size (outputs 7e-5 from the fp32
path against a bound of 1e-2, MPJPE shift 0.001 mm against 0.1 mm).  Plain `bf16` storage (2.9 ms/step) is 4-5e-2 off in train
mode -- a floor any bf16-operand implementation shares, tests/test_bf16_floor_cpu.py shows it on the reference's own operators --
and is reported next to the headline in `other_dtypes` when `--all-dtypes` is given.

Extra objects in that line:
  roofline     -- for the dominant kernel (the MFMA GEMM `gemm_kernel`): algorithmic FLOPs/bytes per launch (SURVEY.md
                  App. C formulas, evaluated per launch from its arguments) / average launch duration measured live with
                  HIP events on the launch stream (hipGraph replay of the step's GEMM launches).
  parity       -- measured in the same run, before the timed region: the timed arithmetic against the fp32 HIP path and against the
                  CPU restatement of the reference (oracle/torch_ops.py) on the same weights and batch, train mode, dropout off.
  cpu_baseline -- the reference's operator sequence restated on stock PyTorch CPU operators (oracle/torch_ops.py, pinned to the
                  reference fixtures; the reference itself cannot travel to the GPU box) timed on this host's cores on a bounded
                  sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import contextlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'gast-net-3dposeestimation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from gast_hip.packer import x3_forward_f16  # noqa: E402

PARENTS17 = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15]   # reference reconstruction.py:95
PARENTS = {17: PARENTS17,
           19: [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 10, 13, 14, 10, 16, 17],      # reference reconstruction.py:87
           15: [-1, 0, 1, 2, 3, 1, 5, 6, 0, 8, 9, 0, 11, 12, 1]}                        # reference common/humaneva_dataset.py:7
# BASELINE.json configs[1..4] (per-GPU batch; configs[3] and [4] quote a global batch over 8 GPUs).  cfg1 is the metric's config.
CONFIGS = {
    'cfg1': dict(J=17, arc=[3, 3, 3], channels=128, batch=128, what='configs[1]: 17 joints, arc 3,3,3 (RF 27), B=128'),
    'cfg2': dict(J=17, arc=[3, 3, 3, 3], channels=64, batch=256, what='configs[2]: 17 joints, arc 3,3,3,3 (RF 81), B=256, channels=64 '
                 '(final width 1024 as in the shipped 81-frame checkpoints)'),
    'cfg3': dict(J=19, arc=[3, 3, 3], channels=128, batch=64, what='configs[3]: 19-joint body+foot, arc 3,3,3, B=512 over 8 GPUs = 64 per GPU'),
    'cfg4': dict(J=15, arc=[3, 3, 3], channels=128, batch=4, what='configs[4]: HumanEva-15, arc 3,3,3, B=32 over 8 GPUs = 4 per GPU -- reported in '
                 'bf16x3: "mixed fp8" is not offered as a mode (e4m3 forward operands through 13 BatchNorm\'d layers move train-mode outputs by 0.4 on a '
                 'range of 1.15; the kernel-level fp8-operand GEMM stays in the library as a tested building block, DESIGN.md section 8)'),
    # not a BASELINE.json config: the shipped 243-frame shape (reference reconstruction.py:225-227, trainval.py -arc 3,3,3,3,3 -ch 32)
    'cfg243': dict(J=17, arc=[3, 3, 3, 3, 3], channels=32, batch=128, what='243-frame model: 17 joints, arc 3,3,3,3,3 (RF 243), channels=32, B=128'),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
def _capture(g, **kw):
    """torch.cuda.graph(g): with a process group alive, RCCL's watchdog thread polls its work events (hipEventQuery) at any time, and in the
    default GLOBAL capture mode such a call from ANOTHER thread is an error -- raised in the watchdog, which then terminates the process
    (seen once in `--force-collective`: 'operation not permitted when stream is capturing').  Thread-local mode only polices the
    capturing thread."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        kw.setdefault('capture_error_mode', 'thread_local')
    return _no_gc_capture(g, kw)


@contextlib.contextmanager
def _no_gc_capture(g, kw):
    """(no cyclic garbage collection inside the capture: a collection may finalize objects of earlier captures whose destructors free
    device memory, which a capturing stream does not allow -- model/gast_net.py: _no_gc)"""
    import gc
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(g, **kw):
            yield
    finally:
        if was:
            gc.enable()


MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'fp32': 157.3, 'bf16x3': 2500.0 / 3, 'fp8': 2500.0}   # dense peaks (bf16x3: three bf16 products per
# FLOP pair; fp8: the forward GEMMs run at the 5 PF fp8 rate, the gradient GEMMs -- 2/3 of the work -- at the bf16 rate: priced at bf16)


def adj_from_parents(parents):
    J = len(parents)
    a = np.zeros((J, J))
    for i, p in enumerate(parents):
        if p >= 0:
            a[i, p] = a[p, i] = 1.0
    a += np.eye(J)
    return torch.from_numpy((a / a.sum(1, keepdims=True)).astype(np.float32))


# --------------------------------------------------------------------------------------------------------- kernel timer
class KernelTimer:
    """Wraps the HipOps methods with HIP-event pairs (recorded on the launch stream) and algorithmic cost models."""

    def __init__(self, ops):
        self.ops = ops
        self.records = []      # (name, start_event, end_event, flops, bytes)
        self.enabled = False
        self.overhead_ms = 0.0
        self.calls = []        # (name, fn, args, kwargs, raw_ms filled in by summary) of the instrumented pass
        for name in ('gemm', 'gemm_multi', 'wgrad', 'wgrad_multi', 'semch_agg_fwd', 'semch_agg_bwd', 'attn_fwd', 'attn_bwd', 'bn_bwd_apply',
                     'bnrelu_apply', 'bnrelu_bwd_mask', 'residual_fwd', 'expand_fwd', 'expand_bwd', 'colsum', 'bn_finalize',
                     'bn_finalize_multi', 'bn_bwd_finalize', 'bn_bwd_finalize_multi', 'bn_bwd_fused_multi', 'semch_adj_fwd_multi',
                     'semch_adj_bwd_multi', 'input_stats', 'adam_step', 'run_pack', 'run_unpack', 'prep'):
            self._wrap(name)

    def calibrate(self, n=200):
        """Duration an event pair reports around an EMPTY kernel on the launch stream (barrier packet, dispatch, barrier
        packet): subtracted from every measurement so that the per-launch figures are kernel durations, comparable with the
        begin/end timestamps rocprofv3 reports."""
        pairs = []
        null = self.ops.null_launch
        for _ in range(n):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            null()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(a.elapsed_time(b) for a, b in pairs)
        self.overhead_ms = v[len(v) // 2]

    def classify(self, name, a, k):
        """which kernel / plan step a gast_gemm(+_multi) call is: 'big' (gemm_big_kernel: the large-M GAST_F32X3 kernel -- the single
        dominant kernel of the step), 'bj' (gemm_bj_kernel: the M = B*J stage's regular shapes, round 6), 'small' (gemm_kernel + split-K
        finish: what is left of that stage), and for the temporal convolution of the
        north star's "conv path": 'conv_fwd' (k taps of one BatchNorm'd tensor as K segments) / 'conv_dgrad' (its input gradient)"""
        try:
            if name == 'gemm_multi':
                jobs = a[0]
                j0 = dict(jobs[0])
                big = self.ops.gemm_path(j0.pop('dom'), j0.pop('N'), j0.pop('segs'), j0.pop('C_'), j0.pop('cmap'), **j0)
                taps = len(jobs) >= 2 and all(len(j['segs']) == 1 and j['segs'][0]['A'].data_ptr() == jobs[0]['segs'][0]['A'].data_ptr()
                                              and j.get('epi', 0) == 2 for j in jobs)
                return ({1: 'big', 2: 'bj'}.get(big, 'small')), ('conv_dgrad' if taps else None)
            big = self.ops.gemm_path(*a, **k)
            segs = a[2]
            same = len(segs) >= 2 and all(sg['A'].data_ptr() == segs[0]['A'].data_ptr() for sg in segs)
            conv = None
            if same and segs[0].get('pro', 0) != 0:
                conv = 'conv_fwd'
            elif same and k.get('addend') is not None:
                conv = 'conv_dgrad'
            return ({1: 'big', 2: 'bj'}.get(big, 'small')), conv
        except Exception:
            return 'small', None

    def replay_only(self, name='gemm', reps=20, pred=None):
        """Average duration of the `name` launches of one step, timed the way they run in the benchmark proper: all of them
        (same arguments and buffers as in the instrumented pass) captured into a hipGraph of their own and replayed `reps` times
        between ONE event pair -- back-to-back dispatch at full clocks.  (The per-launch event pairs of the eager pass leave the
        GPU idle between launches; its kernels run ~20 % slower than in the replayed step.)  Returns ms per launch or None."""
        calls = [r for r in self.calls if r[0] == name and (pred is None or pred(r))]
        if not calls:
            return None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for r in calls:
                    r[1](*r[2], **r[3])
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with _capture(g):
                for r in calls:
                    r[1](*r[2], **r[3])
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (reps * len(calls))
        except Exception:
            return None

    def _wrap(self, name):
        orig = getattr(self.ops, name)
        cost = getattr(self, 'cost_' + name, None)

        def wrapped(*a, **k):
            if not self.enabled:
                return orig(*a, **k)
            fl, by = cost(*a, **k) if cost else (0.0, 0.0)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            self.records.append((name, e0, e1, fl, by))
            if name in ('gemm', 'gemm_multi'):
                kern, conv = self.classify(name, a, k)
                self.calls.append(['gemm', orig, a, k, e0, e1, kern, conv, fl, by])
            return r
        setattr(self.ops, name, wrapped)

    @staticmethod
    def _es(t):
        return t.element_size()

    def cost_gemm(self, dom, N, segs, C_, cmap, bias=None, addend=None, addmap=None, epi=0, partials=None, X=None, **kw):
        B, Tn, J = dom
        M = B * Tn * J
        Ktot = sum(s['K'] for s in segs)
        flops = 2.0 * M * N * Ktot
        es = self._es(segs[0]['A'])
        by = 0.0
        seen = {}
        for s in segs:   # distinct input rows are read once (taps of one tensor overlap)
            key = s['A'].data_ptr()
            rows_tensor = B * s['map'].T_total * J
            seen.setdefault(key, [0, rows_tensor, s['K']])
            seen[key][0] += M
        for cnt, rows_tensor, K in seen.values():
            by += min(cnt, rows_tensor) * K * es
        by += N * Ktot * es                       # weights once
        by += M * N * self._es(C_)                # output once
        if X is not None:
            by += M * N * es
        if addend is not None:
            by += min(M, B * addmap.T_total * J) * N * es
        return flops, by

    def cost_gemm_multi(self, jobs):
        fl = by = 0.0
        for j in jobs:
            j = dict(j)
            f, b = self.cost_gemm(j.pop('dom'), j.pop('N'), j.pop('segs'), j.pop('C_'), j.pop('cmap'), **j)
            fl += f
            by += b
        return fl, by

    def cost_bn_bwd_fused_multi(self, jobs):
        return (sum(4.0 * j['rows'] * j['N'] for j in jobs), sum(3.0 * j['rows'] * j['N'] * self._es(j['dz']) for j in jobs))

    def cost_wgrad(self, dom, P, R, pmap, segs, dW, **kw):
        B, Tn, J = dom
        M = B * Tn * J
        Stot = sum(s['S'] for s in segs)
        flops = 2.0 * M * R * Stot
        es = self._es(P)
        by = M * R * es + R * Stot * 4.0
        seen = {}
        for s in segs:
            key = s['Q'].data_ptr()
            seen.setdefault(key, [0, B * s['map'].T_total * J, s['S']])
            seen[key][0] += M
        for cnt, rows_tensor, S in seen.values():
            by += min(cnt, rows_tensor) * S * es
        return flops, by

    def cost_wgrad_multi(self, jobs):
        fl = by = 0.0
        for j in jobs:
            f, b = self.cost_wgrad(**j)
            fl += f
            by += b
        return fl, by

    def cost_semch_agg_fwd(self, H, F, J, C_, *a, **k):
        return 4.0 * F * J * C_ * 3, F * J * 6.0 * C_ * self._es(H)          # read h0/h1 of both graphs (4C), write Y (2C)

    def cost_semch_agg_bwd(self, dY, H, F, J, C_, *a, **k):
        return 8.0 * F * J * C_ * 3, F * J * 10.0 * C_ * self._es(H)         # read dY (2C) + H (4C), write dH (4C)

    def cost_attn_fwd(self, G, AC, C_k, F, J, C_, nheads, Y):
        return 2.0 * F * J * J * C_, F * J * (2.0 * C_ + 2 * nheads) * self._es(G)

    def cost_attn_bwd(self, dY, G, AC, C_k, F, J, C_, nheads, *a, **k):
        return 4.0 * F * J * J * C_, F * J * (3.0 * C_ + 4 * nheads) * self._es(G)

    def cost_bn_bwd_apply(self, dz, X, rows, N, *a):
        return 4.0 * rows * N, 3.0 * rows * N * self._es(dz)

    def cost_bnrelu_apply(self, X, rows, N, scale, shift, Y, **k):
        return 2.0 * rows * N, 2.0 * rows * N * self._es(X)

    def cost_bnrelu_bwd_mask(self, dY, X, rows, N, *a, **k):
        return 3.0 * rows * N, 3.0 * rows * N * self._es(X)

    def cost_residual_fwd(self, O, omap, scO, shO, T2, sc2, sh2, use_drop, salt, drop, B, Tn, J, C_, Xn):
        return 6.0 * B * Tn * J * C_, 3.0 * B * Tn * J * C_ * self._es(O)

    def cost_expand_fwd(self, x, B, T_in, J, F_in, k0, t_stride, W, sc0, sh0, C_, E, partials, **k):
        rows = E.shape[0]
        return 2.0 * rows * C_ * F_in * k0, rows * C_ * self._es(E) + x.numel() * 4.0

    def cost_expand_bwd(self, dE, x, B, T_in, J, F_in, k0, t_stride, mean0, rstd0, C_, *a, **k):
        rows = dE.shape[0]
        return 2.0 * rows * C_ * (F_in * k0 + 1), rows * C_ * self._es(dE) + x.numel() * 4.0

    def cost_adam_step(self, p, g, m, v, vmax, *a, **k):
        return 12.0 * p.numel(), p.numel() * 4.0 * (9 if vmax is not None else 7)    # read p,g,m,v(,vmax), write p,m,v(,vmax)

    def summary(self):
        agg = {}
        for name, e0, e1, fl, by in self.records:
            ms = max(e0.elapsed_time(e1) - self.overhead_ms, 0.0)
            a = agg.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, roof_ms=0.0, roof_hbm_ms=0.0, roof_mfma_ms=0.0))
            a['launches'] += 1
            a['ms'] += ms
            a['flops'] += fl
            a['bytes'] += by
        return agg


# --------------------------------------------------------------------------------------------------------- cpu baseline
def _stock_steps(device, autocast_bf16, B, budget_s, max_steps, warm_B=None, min_steps=1):
    """Training steps (reference main.py:227-239: zero_grad, forward, mpjpe, backward, Adam amsgrad) of the oracle's
    restatement of the model on STOCK PyTorch operators (oracle/torch_ops.py: F.conv2d / F.batch_norm / matmul / softmax / cat,
    torch autograd) -- the operator sequence the reference issues -- on `device`.  Returns (steps, seconds)."""
    from oracle import gast_oracle as go
    from oracle import torch_ops
    from model.gast_net import SpatioTemporalModel
    adj = adj_from_parents(PARENTS17)
    torch.manual_seed(0)
    m = SpatioTemporalModel(adj, 17, 2, 17, filter_widths=[3, 3, 3], channels=128, dropout=0.05)
    state = {k: v.detach().to(device) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    x = (torch.rand(B, 27, 17, 2, generator=g) * 2 - 1).to(device)
    y = (torch.randn(B, 1, 17, 3, generator=g) * 0.3).to(device)
    y[:, :, 0] = 0
    sync = torch.cuda.synchronize if device != 'cpu' else (lambda: None)
    with go.use_backend(torch_ops):
        om = go.OracleModel(adj.numpy(), [3, 3, 3], 128, dropout=0.05, dtype=torch.float32)
        P = go._P(state, torch.float32)
        opt = torch.optim.Adam([v.v for v in P.leaves.values()], lr=1e-3, amsgrad=True)

        def step(xb, yb):
            opt.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bool(autocast_bf16)):
                pred, _ = om.forward(None, xb, training=True, P=P)
            loss = torch.mean(torch.norm(pred.v.float() - yb, dim=-1))
            loss.backward()
            opt.step()
        wb = warm_B or B
        step(x[:wb], y[:wb])          # warm-up (thread pools / MIOpen find / page faults)
        sync()
        t0 = time.time()
        n = 0
        while n < max_steps and (n < min_steps or time.time() - t0 < budget_s):
            step(x, y)
            n += 1
        sync()
        return n, time.time() - t0


def cpu_baseline(seconds_budget=45.0):
    """The reference's CPU path restated on the same stock PyTorch CPU operators (ATen / oneDNN), all host cores, bounded sample:
    one full-size untimed warm-up step, then 3 timed steps (more only if they fit the budget)."""
    B = 128
    threads = torch.get_num_threads()
    n, dt = _stock_steps('cpu', False, B, seconds_budget, 6, warm_B=B, min_steps=3)
    return dict(value=round(B * n / dt, 3), unit='sequences/s', cores=int(threads), kind='port',
                host_cpus=os.cpu_count(),
                sample='%d full training steps after one full-size warm-up step (zero_grad+fwd+mpjpe+bwd+Adam amsgrad, fp32, B=128 T=27 '
                       'J=17 C=128, dropout 0.05) of the oracle restatement on stock PyTorch CPU operators (oracle/torch_ops.py: the '
                       'ATen operator sequence the reference issues; pinned to the reference fixtures by tests/test_oracle_golden.py), '
                       '%d torch threads; the reference itself cannot travel to this box -- it measured 13.6 seq/s on 8 Xeon cores in '
                       'the build container (BASELINE.md section 2)' % (n, threads))


def cpu_reference_forward(state, adj, fw, channels, x, y3d):
    """Train-mode forward (batch statistics, dropout off) of the CPU restatement of the reference on `state` / `x`: (pred, mpjpe)."""
    from oracle import gast_oracle as go
    from oracle import torch_ops
    with go.use_backend(torch_ops), torch.no_grad():
        om = go.OracleModel(adj.numpy(), list(fw), channels, dropout=0.0, dtype=torch.float32)
        pred, _ = om.forward({k: v.detach().cpu() for k, v in state.items()}, x.cpu(), training=True)
    p = pred.v.float()
    return p, float(torch.mean(torch.norm(p - y3d.cpu(), dim=-1)))


def stock_gpu_baseline(full=False):
    """SURVEY.md section 8(d) "extra comparator": the same model through stock PyTorch-ROCm operators on this GPU (what
    `model.cuda()` of the reference gives a user): eager launches, fp32 and torch.autocast(bfloat16) (3 steps each after a warm-up: part of
    the default N = 1 line; --stock-baseline: a longer sample)."""
    out = {}
    for tag, ac in (('fp32', False), ('autocast_bf16', True)):
        n, dt = _stock_steps('cuda', ac, 128, 10.0 if full else 0.0, 10 if full else 3, min_steps=3)
        out[tag] = dict(ms_per_step=round(dt / n * 1e3, 2), sequences_per_s=round(128 * n / dt, 1), steps=n)
    out['note'] = ('oracle restatement on stock ATen/MIOpen/rocBLAS operators (oracle/torch_ops.py), eager, '
                   'zero_grad+fwd+mpjpe+bwd+Adam(amsgrad), B=128 T=27 J=17 C=128, dropout 0.05')
    return out


# --------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)      # (round 6: 20 -> 100: one 8 ms stall of the box inside 20 steps of 3.4 ms read as 3.80 ms / step)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--dtype', default=os.environ.get('GAST_HIP_DTYPE', 'bf16x3'), choices=['bf16', 'bf16x3', 'fp32', 'f16'])
    ap.add_argument('--variant', default='dilated', choices=['dilated', 'strided'])
    ap.add_argument('--config', default='cfg1', choices=sorted(CONFIGS), help='BASELINE.json configs[1..4] (cfg1 is the metric) or cfg243, the shipped 243-frame shape')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the config\'s)')
    ap.add_argument('--channels', type=int, default=None)
    ap.add_argument('--overlap', action='store_true',
                    help='N > 1: bucketed gradient exchange -- each stage\'s bucket is all-reduced on a communication stream while the '
                         'backward pass of the shallower stages runs (default: one all-reduce between the two graphs of the step)')
    ap.add_argument('--no-parity', action='store_true', help='skip the parity object (timed arithmetic vs fp32 HIP path vs CPU restatement)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-parity', action='store_true', help='with --no-cpu-baseline: still run the CPU restatement FORWARD for the parity object (not its timing)')
    ap.add_argument('--stock-baseline', action='store_true',
                    help='the stock PyTorch-ROCm comparator (SURVEY 8d; fp32, 3 steps by default) also in autocast-bf16 and with a longer sample')
    ap.add_argument('--no-stock-baseline', action='store_true', help='skip the stock PyTorch-ROCm comparator')
    ap.add_argument('--no-graph', action='store_true', help='run the step eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--timer-steps', type=int, default=3, help='eager, event-instrumented steps for the per-kernel roofline')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--no-eager', action='store_true', help='skip the eager-launch timing of the same step (profiling runs: keeps the trace tail on the replayed graph)')
    ap.add_argument('--split-graph', action='store_true', help='force the two-graph step (the default for --gpus > 1)')
    ap.add_argument('--full-graph', action='store_true', help='--gpus > 1: capture the RCCL all-reduce inside one graph')
    ap.add_argument('--force-collective', action='store_true',
                    help='single process: still create the RCCL process group and run the all-reduce / barriers of the N > 1 path '
                         '(exercises that code path on a 1-GPU box)')
    ap.add_argument('--torch-tail', action='store_true', help='eager mpjpe formula + torch fused Adam instead of the HIP loss/optimizer')
    ap.add_argument('--no-f16', action='store_true', help='skip the GAST_HIP_DTYPE=f16 variant (binary16 storage, the 16-bit mode) in the N = 1 line')
    ap.add_argument('--no-twin', action='store_true', help='skip the second model variant (SpatioTemporalModelOptimized1f) in the N = 1 line')
    ap.add_argument('--dry-run-cpu', action='store_true',
                    help='LAUNCHER DRY RUN, not a measurement: the same self-launch / rendezvous / per-rank seeds / (bucketed) gradient exchange / '
                         'single JSON line on rank 0, over the gloo backend with the kernels replaced by the test-suite\'s numpy op mirror '
                         '(tests/fake_backend.py); `value` is null.  Exists so that the first multi-GPU run is not the first execution of '
                         'that code (tests/test_dist_cpu.py)')
    args = ap.parse_args()
    dry = args.dry_run_cpu
    if dry:
        args.no_graph = args.no_parity = args.no_cpu_baseline = args.no_kernel_timer = args.no_eager = args.no_twin = args.no_f16 = args.no_stock_baseline = True
        args.torch_tail = True        # (the fused loss / flat Adam are HIP launches without a CPU form)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if dry:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
    elif args.gpus > 1:
        # not under torchrun: start the N ranks ourselves (one process per GPU, RCCL over xGMI) and relay rank 0's JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr',
               '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    dev = torch.device('cpu') if dry else torch.device('cuda', local_rank)
    if dry:
        args.dtype = 'fp32'
        sys.path.insert(0, os.path.join(ROOT, 'tests'))

        def _sync():
            pass
    else:
        _sync = torch.cuda.synchronize
    os.environ['GAST_HIP_DTYPE'] = args.dtype
    cfg = CONFIGS[args.config]
    J, arc = cfg['J'], cfg['arc']
    C = args.channels or cfg['channels']
    B = args.batch or cfg['batch']
    T = int(np.prod(arc))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()          # before touching the GPU (always on the metric's config, configs[1])

    from model.gast_net import SpatioTemporalModel, SpatioTemporalModelOptimized1f
    from gast_hip.dist import FlatGradAllReduce
    torch.manual_seed(0)
    cls = SpatioTemporalModel if args.variant == 'dilated' else SpatioTemporalModelOptimized1f
    adj = adj_from_parents(PARENTS[J])
    model = cls(adj, J, 2, J, filter_widths=arc, causal=False, dropout=0.05, channels=C).to(dev)
    model._runner.graph_mode = False      # (the module's own hipGraph replay is measured explicitly in the `module_graph` leg below)
    if dry:
        from fake_backend import use_oracle_ops      # (dry run only: the host plan on the numpy mirror of the op set)
        use_oracle_ops(model)
    model.train()
    g = torch.Generator().manual_seed(1234 + rank)     # reference generator seed (common/generators.py:26), per-rank shard
    x = (torch.rand(B, T, J, 2, generator=g) * 2 - 1).to(dev)
    y3d = torch.randn(B, 1, J, 3, generator=g) * 0.3
    y3d[:, :, 0] = 0                                     # reference main.py:225
    y3d = y3d.to(dev)

    # ---- parity, measured in this run before anything is timed: the same weights and batch, train mode (batch statistics), dropout
    # off, through (a) the timed arithmetic, (b) the fp32 HIP path, (c) the CPU restatement of the reference on stock ATen operators
    parity = None
    cpu_ref = None
    if rank == 0 and not args.no_parity:
        try:
            pm = cls(adj, J, 2, J, filter_widths=arc, causal=False, dropout=0.0, channels=C)
            pm._runner.graph_mode = False
            pm.load_state_dict(model.state_dict())
            pm.to(dev).train()
            sd = {k: v.clone() for k, v in pm.state_dict().items()}
            outs = {}
            for dt_ in dict.fromkeys((args.dtype, 'fp32')):
                os.environ['GAST_HIP_DTYPE'] = dt_
                pm.load_state_dict(sd)
                with torch.no_grad():
                    yy = pm(x).float()
                outs[dt_] = (yy, float(torch.mean(torch.norm(yy - y3d, dim=-1))))
            os.environ['GAST_HIP_DTYPE'] = args.dtype
            d32 = float((outs[args.dtype][0] - outs['fp32'][0]).abs().max())
            parity = {'mode': 'train-mode forward (batch-statistic BatchNorm), dropout off, same weights and batch as the timed step',
                      'dtype': args.dtype, 'output_abs_max': round(float(outs['fp32'][0].abs().max()), 4),
                      'vs_fp32_hip': {'max_abs': d32, 'mpjpe_shift_mm': abs(outs[args.dtype][1] - outs['fp32'][1]) * 1e3},
                      # bf16x3 is fp32 storage with fp32-class products: it is gated at the north star's FP32 bound (forward GEMMs on
                      # fp16 hi/lo pairs: measured 1.1e-5 .. 2.8e-5 over the BASELINE configs), NOT at the 1e-2 the north star grants
                      # bf16; with GAST_X3_FWD=bf16 (bf16 pairs in the forward too: 7e-5 .. 2.4e-4) at 2e-4 x 2^(levels - 3)
                      'tolerance': {'max_abs': {'fp32': 1e-4, 'bf16x3': 1e-4 if x3_forward_f16() else 2e-4 * 2 ** max(0, len(arc) - 3)}.get(args.dtype, 1e-2),
                                    'mpjpe_mm': 0.1,
                                    'source': 'BASELINE.json north_star: 1e-4 fp32 / 1e-2 bf16, MPJPE within 0.1 mm; bf16x3 '
                                              '(fp32 storage, hi/lo split products) is held to the fp32 bound'}}
            if world == 1 and (not args.no_cpu_baseline or args.cpu_parity):
                yc, lc = cpu_reference_forward(sd, adj, arc, C, x, y3d)
                cpu_ref = (yc, lc)             # (the 16-bit variant's parity block below compares with the same restatement)
                dc = float((outs[args.dtype][0].cpu() - yc).abs().max())
                parity['vs_cpu_reference_restatement'] = {'max_abs': dc, 'mpjpe_shift_mm': abs(outs[args.dtype][1] - lc) * 1e3,
                                                          'fp32_hip_max_abs': float((outs['fp32'][0].cpu() - yc).abs().max()),
                                                          'what': 'oracle/torch_ops.py on stock PyTorch CPU operators, fp32'}
                d32 = max(d32, dc)
            parity['pass'] = bool(d32 < parity['tolerance']['max_abs'] and parity['vs_fp32_hip']['mpjpe_shift_mm'] < 0.1)
            del pm, outs
        except Exception as e:     # noqa: BLE001 -- never lose the bench line to the parity leg
            parity = {'error': str(e).splitlines()[0][:200]}

    sync = FlatGradAllReduce(model.parameters(), model=model, buckets=3 if (args.overlap and len(arc) >= 2) else 1)
    sync.force = args.force_collective
    use_graph = not args.no_graph
    if args.torch_tail:   # the eager-formula loss and torch's fused multi-tensor Adam (comparison only)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=True, **({} if dry else dict(fused=True, capturable=use_graph)))
        loss_fn = lambda p, t: torch.mean(torch.norm(p - t, dim=-1))    # noqa: E731 -- mpjpe, reference common/loss.py:5-11
    else:                 # reference trainval.py:78 Adam(amsgrad=True) / common/loss.py mpjpe as single HIP launches (row f1)
        from gast_hip.optim import FlatAdam
        from gast_hip.loss import mpjpe as loss_fn
        opt = FlatAdam(model.parameters(), lr=1e-3, amsgrad=True, ops=model._runner.engine.ops)
        sync.attach(opt)          # the 1/world averaging rides in the Adam kernel's gradient scale

    timer = None
    if not args.no_kernel_timer and rank == 0:
        timer = KernelTimer(model._runner.engine.ops)

    def step():
        sync.zero_(defer=True)     # (rides in the forward's pass prologue: one launch with the arena fill and the seed bump)
        pred = model(x)
        loss = loss_fn(pred, y3d)
        loss.backward()
        sync.sync()
        opt.step()
        return loss

    def step_compute():      # everything before the gradient exchange
        sync.zero_(defer=True)     # (rides in the forward's pass prologue: one launch with the arena fill and the seed bump)
        pred = model(x)
        loss = loss_fn(pred, y3d)
        loss.backward()
        return loss

    # ---- untimed warm-up (also the side-stream warm-up torch requires before capture), then capture the step.
    # 'full': ONE hipGraph for the whole step (single GPU; with --full-graph also across the RCCL all-reduce).
    # 'split' (default for N > 1): graph A = zero_grad + forward + loss + backward, the flat-buffer all-reduce launched eagerly on
    # the same stream, graph B = Adam -- no collective inside a captured graph, two graph launches + one RCCL call per step.
    mode, graph_note = 'eager', 'eager'
    graphs, static_loss = [], None
    want = 'eager' if not use_graph else ('split' if (args.split_graph or (collective and not (args.full_graph or args.overlap))) else 'full')
    if want != 'eager':
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(3, args.warmup)):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if want == 'full':
                g0 = torch.cuda.CUDAGraph()
                with _capture(g0):
                    static_loss = step()
                graphs = [g0]
                graph_note = 'hipGraph replay of the whole step (captured through torch.cuda.graph)'
            else:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with _capture(ga):
                    static_loss = step_compute()
                with _capture(gb, pool=ga.pool()):
                    opt.step()
                graphs = [ga, gb]
                graph_note = ('two hipGraphs per step (zero_grad+fwd+loss+bwd | Adam) around an eagerly launched flat-gradient '
                              'all-reduce')
            mode = want
        except Exception as e:   # noqa: BLE001 -- report and fall back to eager launches (same kernels)
            graphs = []
            graph_note = 'eager (graph capture failed: %s)' % (str(e).splitlines()[0][:120],)
            torch.cuda.synchronize()

    def run_step():
        if mode == 'full':
            graphs[0].replay()
        elif mode == 'split':
            graphs[0].replay()
            sync.sync()
            graphs[1].replay()
        else:
            return step()
        return static_loss

    def barrier():
        if collective:
            dist.barrier() if dry else dist.barrier(device_ids=[local_rank])

    for _ in range(args.warmup):
        run_step()
    _sync()
    # host hygiene before the timed region: a generation-2 pass of Python's cyclic collector over the module / tensor graph costs
    # milliseconds; everything alive now is moved to the permanent generation (what a long training loop reaches by itself)
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    _sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step()
    _sync()
    barrier()
    _sync()
    elapsed = time.perf_counter() - t0
    gc.unfreeze()
    per_rank_ms = None
    if collective:
        tt = torch.zeros(world, dtype=torch.float64, device=dev)
        tt[rank] = elapsed
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(v) / args.steps * 1e3, 4) for v in tt.tolist()]
        elapsed = float(tt.max().item())
    # ---- the same step launched eagerly (what an unchanged training script gets: it calls model(x) / loss.backward() /
    # optimizer.step() kernel by kernel through ctypes; the hipGraph is built by this benchmark only)
    eager_ms = module_graph_ms = None
    if rank == 0 and world == 1 and mode != 'eager' and not args.no_eager:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - te) / 10 * 1e3
        # ... and with GAST_HIP_GRAPH=1: the module itself replays its forward / backward from hipGraphs it captured on the third
        # call (model(x) and loss.backward() of an unchanged loop), the loss / optimizer launches stay eager
        try:
            model._runner.graph_mode = True
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            te = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            module_graph_ms = (time.perf_counter() - te) / 10 * 1e3
        except Exception as e:   # noqa: BLE001 -- never lose the bench line to this leg
            module_graph_ms = None
            print('module graph leg failed: %s' % (str(e).splitlines()[0][:200],), file=sys.stderr)
        finally:
            model._runner.graph_mode = False
            model._runner._graphs.clear()
    # ---- per-kernel durations: the same step, eagerly, with a HIP-event pair around every launch (events cannot be
    # recorded inside a replayed graph; the kernels and their arguments are identical to the replayed ones)
    if timer and rank == 0:
        timer.calibrate()
        timer.enabled = True
        for _ in range(max(1, args.timer_steps)):
            step_compute()       # rank 0 only: no collective in the instrumented pass
            opt.step()
        torch.cuda.synchronize()
        timer.enabled = False
        gemm_replay_ms = timer.replay_only('gemm')
        tsteps = max(1, args.timer_steps)

    # ---- forward pass alone (the north star quotes its roofline target on the forward): train-mode forward under no_grad,
    # captured into its own hipGraph and replayed 20x inside one event pair
    fwd_ms = None
    if rank == 0 and world == 1 and use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(3):
                    model(x)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gf = torch.cuda.CUDAGraph()
            with _capture(gf), torch.no_grad():
                model(x)
            for _ in range(3):
                gf.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                gf.replay()
            e1.record()
            torch.cuda.synchronize()
            fwd_ms = e0.elapsed_time(e1) / 20
        except Exception:
            fwd_ms = None

    # ---- "report both model variants".  The headline (`value`) is the north-star-named dilated SpatioTemporalModel;
    # the twin the reference actually trains with at stride 1 (SpatioTemporalModelOptimized1f, reference main.py:166-171) is timed here in
    # the same run, same config / batch / arithmetic / step definition / steps / warm-up, as its own replayed hipGraph.
    twin = None
    if rank == 0 and world == 1 and not args.no_twin and use_graph and mode == 'full' and not args.torch_tail:
        try:
            tcls = SpatioTemporalModelOptimized1f if args.variant == 'dilated' else SpatioTemporalModel
            torch.manual_seed(0)
            tm = tcls(adj, J, 2, J, filter_widths=arc, causal=False, dropout=0.05, channels=C).to(dev).train()
            tm._runner.graph_mode = False
            tsync = FlatGradAllReduce(tm.parameters(), model=tm, buckets=1)
            topt = FlatAdam(tm.parameters(), lr=1e-3, amsgrad=True, ops=tm._runner.engine.ops)
            tsync.attach(topt)

            def tstep():
                tsync.zero_()
                l_ = loss_fn(tm(x), y3d)
                l_.backward()
                tsync.sync()
                topt.step()
                return l_
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    tstep()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            tg = torch.cuda.CUDAGraph()
            with _capture(tg):
                tloss = tstep()
            for _ in range(args.warmup):
                tg.replay()
            torch.cuda.synchronize()
            tt0 = time.perf_counter()
            for _ in range(args.steps):
                tg.replay()
            torch.cuda.synchronize()
            tms = (time.perf_counter() - tt0) / args.steps * 1e3
            twin = {'variant': 'strided' if args.variant == 'dilated' else 'dilated', 'model': tcls.__name__, 'ms_per_step': round(tms, 4),
                    'sequences_per_s': round(B / tms * 1e3, 1), 'steps': args.steps, 'warmup': args.warmup, 'loss_last': round(float(tloss.item()), 6),
                    'note': 'same config, batch, arithmetic and step definition as `value`, own hipGraph; the strided twin does 0.31x the '
                            'FLOPs of the dilated model on T = RF windows (SURVEY.md App. C) and is what reference main.py:166-171 trains'}
            del tm, tsync, topt, tg
        except Exception as e:   # noqa: BLE001 -- never lose the bench line to this leg
            twin = {'error': str(e).splitlines()[0][:200]}

    # ---- the 16-bit mode (BASELINE.json configs[1] says "bf16 forward+backward", north_star grants 16-bit arithmetic 1e-2): the same
    # model, batch and step in GAST_HIP_DTYPE=f16 -- IEEE binary16 storage and matrix operands (libgast_hip_f16.so), fp32 accumulate /
    # statistics / master weights, loss-scaled gradients -- with ITS OWN parity block against the 16-bit bound.  bfloat16 cannot meet
    # that bound in train mode (tests/test_bf16_floor_cpu.py); binary16's 11 significand bits do.  `value` stays the fp32-class mode.
    f16 = None
    if rank == 0 and world == 1 and not args.no_f16 and use_graph and mode == 'full' and not args.torch_tail and args.dtype != 'f16':
        prev_dt = os.environ.get('GAST_HIP_DTYPE')
        try:
            torch.manual_seed(0)
            fm = cls(adj, J, 2, J, filter_widths=arc, causal=False, dropout=0.05, channels=C).to(dev)
            fm._runner.graph_mode = False
            f16 = {}
            if not args.no_parity:
                pm = cls(adj, J, 2, J, filter_widths=arc, causal=False, dropout=0.0, channels=C)
                pm._runner.graph_mode = False
                pm.load_state_dict(fm.state_dict())
                pm.to(dev).train()
                sd = {k: v.clone() for k, v in pm.state_dict().items()}
                outs = {}
                for dt_ in ('f16', 'fp32'):
                    os.environ['GAST_HIP_DTYPE'] = dt_
                    pm.load_state_dict(sd)
                    with torch.no_grad():
                        yy = pm(x).float()
                    outs[dt_] = (yy, float(torch.mean(torch.norm(yy - y3d, dim=-1))))
                d16 = float((outs['f16'][0] - outs['fp32'][0]).abs().max())
                sh16 = abs(outs['f16'][1] - outs['fp32'][1]) * 1e3
                f16['parity'] = {'mode': 'train-mode forward (batch-statistic BatchNorm), dropout off, vs the fp32 HIP path',
                                 'output_abs_max': round(float(outs['fp32'][0].abs().max()), 4), 'max_abs': d16, 'mpjpe_shift_mm': sh16,
                                 'tolerance': {'max_abs': 1e-2, 'mpjpe_mm': 0.1, 'source': 'BASELINE.json north_star: 1e-2 for 16-bit arithmetic, '
                                                                                         'MPJPE within 0.1 mm'},
                                 'pass': bool(d16 < 1e-2 and sh16 < 0.1)}
                if cpu_ref is not None:        # oracle side: the CPU restatement of the reference (same weights and batch: same seeds)
                    dc16 = float((outs['f16'][0].cpu() - cpu_ref[0]).abs().max())
                    f16['parity']['vs_cpu_reference_restatement'] = {'max_abs': dc16, 'mpjpe_shift_mm': abs(outs['f16'][1] - cpu_ref[1]) * 1e3,
                                                                     'what': 'oracle/torch_ops.py on stock PyTorch CPU operators, fp32'}
                    f16['parity']['pass'] = bool(f16['parity']['pass'] and dc16 < 1e-2 and abs(outs['f16'][1] - cpu_ref[1]) * 1e3 < 0.1)
                f16['parity']['scope'] = ('specified for BASELINE configs[1] and [3]; at configs[2] (C0 = 64, four temporal levels, B = 256) the mode '
                                          'sits AT the 1e-2 bound (9.7e-3 .. 9.8e-3 measured): tests/test_f16_gpu.py')
                del pm, outs
            os.environ['GAST_HIP_DTYPE'] = 'f16'
            fm.train()
            fsync = FlatGradAllReduce(fm.parameters(), model=fm, buckets=1)
            fopt = FlatAdam(fm.parameters(), lr=1e-3, amsgrad=True, ops=fm._runner.engine.ops)
            fsync.attach(fopt)

            def fstep():
                fsync.zero_(defer=True)
                l_ = loss_fn(fm(x), y3d)
                l_.backward()
                fsync.sync()
                fopt.step()
                return l_
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fstep()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            fg = torch.cuda.CUDAGraph()
            with _capture(fg):
                floss = fstep()
            for _ in range(args.warmup):
                fg.replay()
            torch.cuda.synchronize()
            ft0 = time.perf_counter()
            for _ in range(args.steps):
                fg.replay()
            torch.cuda.synchronize()
            fms = (time.perf_counter() - ft0) / args.steps * 1e3
            f16.update({'dtype': 'f16', 'ms_per_step': round(fms, 4), 'sequences_per_s': round(B / fms * 1e3, 1), 'steps': args.steps,
                        'warmup': args.warmup, 'loss_last': round(float(floss.item()), 6),
                        'arithmetic': 'IEEE binary16 storage and MFMA operands (v_mfma_f32_32x32x16_f16), fp32 accumulate / statistics / '
                                      'softmax / master weights / parameter gradients; activation gradients travel x%g (loss scale)'
                                      % float(os.environ.get('GAST_F16_LOSS_SCALE', '4096')),
                        'note': 'same model, batch and step as `value`, own hipGraph'})
            # the 16-bit counterpart of `forward_only` (north_star: >= 60 % of the roofline on the forward): train-mode forward under
            # no_grad, own hipGraph, 20 replays in one event pair; roofline from the 16-bit bytes of SURVEY.md App. C
            try:
                with torch.no_grad():
                    for _ in range(3):
                        fm(x)
                    torch.cuda.synchronize()
                    fgf = torch.cuda.CUDAGraph()
                    with _capture(fgf):
                        fm(x)
                    for _ in range(3):
                        fgf.replay()
                    fe0, fe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    fe0.record()
                    for _ in range(20):
                        fgf.replay()
                    fe1.record()
                    torch.cuda.synchronize()
                ffwd = fe0.elapsed_time(fe1) / 20
                if args.config == 'cfg1' and C == 128:
                    froof = max(793.4e6 * (B / 128.0) / (HBM_PEAK_GBS * 1e9), 160.7e9 * (B / 128.0) / (MFMA_PEAK_TFLOPS['f16'] * 1e12)) * 1e3
                    f16['forward_only'] = {'ms': round(ffwd, 4), 'sequences_per_s': round(B / ffwd * 1e3, 1), 'roofline_ms': round(froof, 4),
                                           'frac_of_roofline': round(froof / ffwd, 4),
                                           'note': 'train-mode forward of the 16-bit mode, own hipGraph; roofline = max(HBM, MFMA) of the '
                                                   '16-bit algorithmic work of SURVEY.md App. C (793.4 MB, 160.7 GFLOP per B = 128 forward)'}
                    f16['roofline'] = {'bound': 'hbm', 'achieved': round(793.4e6 * (B / 128.0) / (ffwd * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                       'frac': round(793.4e6 * (B / 128.0) / (ffwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': None,
                                       'what': 'whole forward pass (all kernels), algorithmic 16-bit bytes / measured forward time'}
                del fgf
            except Exception as e:   # noqa: BLE001
                f16['forward_only'] = {'error': str(e).splitlines()[0][:160]}
            del fm, fsync, fopt, fg
        except Exception as e:   # noqa: BLE001 -- never lose the bench line to this leg
            f16 = {'error': str(e).splitlines()[0][:200]}
        finally:
            if prev_dt is None:
                os.environ.pop('GAST_HIP_DTYPE', None)
            else:
                os.environ['GAST_HIP_DTYPE'] = prev_dt

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        out = {
            'metric': 'sequences/sec (B=%d, T=%d, J=%d) fwd+bwd' % (B, T, J), 'value': None if dry else round(value, 1), 'unit': 'sequences/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic' if not dry else 'DRY RUN of the launcher on CPU (gloo, numpy op mirror): not a measurement',
            'config': {'workload': '%s%s; SpatioTemporalModel J=%d arc %s (RF %d) channels=%d, '
                                   'B=%d/GPU x T=%d, dropout 0.05, step = zero_grad+fwd+mpjpe+bwd%s+Adam(amsgrad)%s'
                                   % ('BASELINE.json ' if cfg['what'].startswith('configs[') else '', cfg['what'], J, ','.join(map(str, arc)), T, C, B, T,
                                      '+RCCL grad all-reduce' if world > 1 else '', ' [torch loss/optimizer]' if args.torch_tail else ''),
                       'variant': args.variant, 'global_batch': world * B, 'parallelism': 'dp%d' % world,
                       'arithmetic': {'bf16x3': 'fp32 storage; every GEMM product as hi*hi + hi*lo + lo*hi of 16-bit hi/lo pairs, fp32 accumulate: '
                                                + ('forward GEMMs on fp16 pairs (v_mfma_f32_32x32x16_f16, GAST_F32X3H), ' if x3_forward_f16() else
                                                   'forward GEMMs (GAST_X3_FWD=bf16), ')
                                                + 'input / weight gradients on bf16 pairs (v_mfma_f32_32x32x16_bf16, GAST_F32X3)',
                                      'bf16': 'bf16 storage and MFMA operands, fp32 accumulate / statistics / master weights',
                                      'fp32': 'fp32 storage, v_mfma_f32_32x32x2_f32',
                                      'f16': 'IEEE binary16 storage and MFMA operands (v_mfma_f32_32x32x16_f16), fp32 accumulate / statistics / '
                                             'master weights; loss-scaled activation gradients',
                                      'fp8': 'mixed fp8 (BASELINE.json configs[4]): bf16 storage; forward channel GEMMs with OCP e4m3 operands '
                                             '(per-tensor power-of-two weight scales) on v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulate; input / '
                                             'weight gradients, statistics and softmax as in bf16 mode'}[args.dtype],
                       'loss_last': round(float(loss.item()), 6), 'launch': graph_note,
                       'rccl_world_size': world if collective else None,
                       'collective_backend': None if not collective else ('gloo (dry run)' if dry else 'nccl (= RCCL over xGMI)'),
                       'per_rank_input_seeds': [1234 + r for r in range(world)],
                       'gradient_exchange': (None if not collective else {
                           'what': ('%d buckets of the flat fp32 gradient buffer, each all-reduced on a communication stream as soon as its '
                                    'stage of the backward pass is done' % len(sync.ranges)) if len(sync.ranges) > 1
                                   else 'one all-reduce(sum) of the flat fp32 gradient buffer after backward',
                           'bytes_per_step': sync.nbytes(), 'buckets': len(sync.ranges),
                           'bucket_bytes': [4 * (b_ - a_) for a_, b_ in sync.ranges],
                           'overlap_with_backward': len(sync.ranges) > 1,
                           'averaging': '1/world folded into the Adam kernel (grad_scale)' if sync.scale_in_optimizer else 'in-place scale after the all-reduce'})},
        }
        if dry:
            out['dry_run'] = True
        if world > 1:
            out['cpu_baseline'] = None
            out['cpu_baseline_note'] = 'reported by the N = 1 run only (rank 0 at N = 1 times the CPU restatement; see BENCH line of --gpus 1)'
        if eager_ms is not None:
            if module_graph_ms is not None:
                out['module_graph'] = {'ms_per_step': round(module_graph_ms, 4), 'sequences_per_s': round(B / module_graph_ms * 1e3, 1),
                                       'note': 'what an UNCHANGED training loop gets (the default since round 3; GAST_HIP_GRAPH=0 turns it off): '
                                               'model(x) / loss.backward() replayed from the hipGraphs the module captures by itself on the third '
                                               'call of a shape, loss and optimizer launched eagerly'}
            out['eager_launch'] = {'ms_per_step': round(eager_ms, 4), 'sequences_per_s': round(B / eager_ms * 1e3, 1),
                                   'note': 'the same step with every kernel launched eagerly from Python (ctypes; GAST_HIP_GRAPH=0); `value` '
                                           'replays the whole step as one hipGraph'}
        if twin is not None or f16 is not None:
            out['variants'] = {args.variant: {'ms_per_step': round(ms, 4), 'sequences_per_s': round(value, 1), 'headline': True}}
            if twin is not None:
                out['variants'][twin.get('variant', 'twin')] = twin
            if f16 is not None:
                out['variants']['f16'] = f16
        if collective:
            out['per_rank_ms_per_step'] = per_rank_ms
        if parity is not None:
            out['parity'] = parity
        if timer:
            agg = timer.summary()
            peak_tf = MFMA_PEAK_TFLOPS[args.dtype]
            for a in agg.values():
                a['roof_hbm_ms'] = a['bytes'] / (HBM_PEAK_GBS * 1e9) * 1e3
                a['roof_mfma_ms'] = a['flops'] / (peak_tf * 1e12) * 1e3
                a['roof_ms'] = max(a['roof_hbm_ms'], a['roof_mfma_ms'])
            gm = agg.get('gemm')
            if gm and 'gemm_multi' in agg:      # the paired thin GEMMs run as one grid: same kernel family
                gm = {k: gm[k] + agg['gemm_multi'][k] for k in gm}
            if gm and gm['ms'] > 0:
                bound = 'hbm' if gm['roof_hbm_ms'] >= gm['roof_mfma_ms'] else 'mfma'
                eager_avg_ms = gm['ms'] / gm['launches']
                avg_ms = gemm_replay_ms if gemm_replay_ms else eager_avg_ms
                if bound == 'hbm':
                    ach = gm['bytes'] / gm['launches'] / (avg_ms * 1e-3) / 1e9
                    peak, unit = HBM_PEAK_GBS, 'GB/s'
                else:
                    ach = gm['flops'] / gm['launches'] / (avg_ms * 1e-3) / 1e12
                    peak, unit = peak_tf, 'TFLOP/s'
                traffic, tsrc = None, None
                try:     # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, gfx950-corrected)
                    if args.config != 'cfg1' or args.variant != 'dilated':
                        raise LookupError('the committed counter passes are of the configs[1] workload: traffic stays null for any other')
                    pmc_name = next(n for n in ('r06_pmc_hbm_bytes_%s.json' % args.dtype, 'r05_pmc_hbm_bytes_%s.json' % args.dtype, 'r04_pmc_hbm_bytes_%s.json' % args.dtype, 'r03_pmc_hbm_bytes_%s.json' % args.dtype, 'r02_pmc_hbm_bytes_%s.json' % args.dtype)
                                    if os.path.exists(os.path.join(ROOT, 'profiles', n)))      # counters of the committed kernels, newest round first
                    pmc = json.load(open(os.path.join(ROOT, 'profiles', pmc_name)))
                    GEMM_K = ('gemm_kernel<', 'gemm_multi_kernel<', 'gemm_big_kernel<', 'gemm_big_multi_kernel<', 'splitk_finish_kernel<',
                              'splitk_finish_multi_kernel<', 'gemm_bj_kernel<', 'gemm_bj_multi_kernel<')
                    fam = [v for k, v in pmc.items() if k.startswith(GEMM_K)]
                    # bytes of the family per step / gast_gemm(+_multi) API launches per step (a multi call may be two grids)
                    calls = pmc['_meta']['steps'] * gm['launches'] / tsteps
                    traffic = round(sum(v['hbm_bytes_per_launch'] * v['launches'] for v in fam) / calls)
                    tsrc = ('profiles/%s: sum of (2*FETCH_SIZE + WRITE_SIZE) KiB over the gemm / gemm_multi / splitk_finish '
                            'kernels / %d gast_gemm launches, two rocprofv3 --pmc passes of `bench.py --dtype %s --no-graph`' % (pmc_name, calls, args.dtype))
                except Exception:
                    pass
                # ---- the family split by kernel, and the temporal-convolution launches of the north star's "conv path", each timed the same
                # way (its launches alone in a hipGraph, 20 replays in one event pair)
                def klass(pred, label):
                    rs = [r for r in timer.calls if pred(r)]
                    if not rs:
                        return None
                    msl = timer.replay_only('gemm', pred=pred)
                    if not msl:
                        return None
                    nl = len(rs)
                    byt, flo = sum(r[9] for r in rs) / nl, sum(r[8] for r in rs) / nl
                    mult = 3.0 if args.dtype == 'bf16x3' else 1.0
                    nmulti = sum(1 for r in rs if r[0] == 'gemm_multi')
                    return {'what': label, 'launches_per_step': nl / tsteps, 'multi_job_calls_per_step': nmulti / tsteps,
                            'grids_note': 'launches = gast_gemm / gast_gemm_multi API calls; a multi-job call is one grid (gemm_big_multi_kernel / '
                                          'gemm_multi_kernel in a kernel trace; one more grid per extra epilogue variant among its jobs, and '
                                          'one shared split-K finish grid)',
                            'avg_launch_us': round(msl * 1e3, 2),
                            'alg_mb_per_launch': round(byt / 1e6, 3), 'achieved_gb_s': round(byt / (msl * 1e-3) / 1e9, 1),
                            'frac_of_hbm_peak': round(byt / (msl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            'mfma_tflops_eq': round(flo * mult / (msl * 1e-3) / 1e12, 1),
                            'frac_of_mfma_peak': round(flo * mult / (msl * 1e-3) / 1e12 / (2500.0 if args.dtype != 'fp32' else 157.0), 4)}
                by_kernel = {}
                for key, pred, label in (('gemm_big_kernel', lambda r: r[6] == 'big', 'the large-M GAST_F32X3 kernel (csrc/gemm_big.hip): the single dominant kernel of the step'),
                                         ('gemm_bj_kernel', lambda r: r[6] == 'bj', 'the M = B*J stage on the in-block split-K kernel (csrc/gemm_bj.hip, round 6): every shape of the stage whose K steps come in fours'),
                                         ('gemm_kernel+splitk_finish', lambda r: r[6] == 'small', 'what stays on csrc/gemm.hip (split-K + finish): K = 8, K = 5C + 8 + 2C, N = 5C + 8 of the M = B*J stage'),
                                         ('temporal_conv_fwd', lambda r: r[7] == 'conv_fwd', 'forward dilated temporal convolutions (k taps of one BatchNorm\'d tensor as K segments; reference gast_net.py:173)'),
                                         ('temporal_conv_dgrad', lambda r: r[7] == 'conv_dgrad', 'their input gradients (gather GEMM / disjoint-tap scatter GEMMs)')):
                    v = klass(pred, label)
                    if v:
                        by_kernel[key] = v
                if 'gemm_big_kernel' in by_kernel and args.dtype == 'bf16x3':
                    # the vendor bar: ONE plain bf16 product of the kernel's largest shape through torch.mm (hipBLASLt / rocBLAS),
                    # no prologue / statistics / fp32 I/O -- gemm_big does THREE such products plus those per launch
                    try:
                        big = max((r for r in timer.calls if r[6] == 'big' and r[0] == 'gemm'), key=lambda r: r[8])
                        dom_, N_, segs_ = big[2][0], big[2][1], big[2][2]
                        M_, K_ = dom_[0] * dom_[1] * dom_[2], sum(sg['K'] for sg in segs_)
                        Av = torch.randn(M_, K_, device=dev, dtype=torch.bfloat16)
                        Wv = torch.randn(N_, K_, device=dev, dtype=torch.bfloat16)
                        for _ in range(3):
                            Av @ Wv.t()
                        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ev0.record()
                        for _ in range(20):
                            Av @ Wv.t()
                        ev1.record()
                        torch.cuda.synchronize()
                        vus = ev0.elapsed_time(ev1) / 20 * 1e3
                        ours = timer.replay_only('gemm', pred=lambda r: r is big)
                        by_kernel['gemm_big_kernel'].update({'vendor_bf16_us': round(vus, 1), 'vendor_shape': 'M=%d N=%d K=%d' % (M_, N_, K_),
                                                             'this_kernel_us_same_shape': round(ours * 1e3, 1) if ours else None,
                                                             'x3_over_vendor': round(ours * 1e3 / vus, 2) if ours else None,
                                                             'vendor_note': 'torch.mm in bf16 (hipBLASLt/rocBLAS), ONE product, no prologue / epilogue; '
                                                                            'this kernel: three split products + BN prologue + statistics + fp32 I/O'})
                        del Av, Wv
                    except Exception as e:      # noqa: BLE001
                        by_kernel['gemm_big_kernel']['vendor_error'] = str(e).splitlines()[0][:160]
                out['roofline_by_kernel'] = by_kernel
                out['roofline'] = {'kernel': ('gemm_big_kernel (large-M GAST_F32X3) + gemm_kernel / gemm_multi_kernel<%s> (gast_gemm, gast_gemm_multi; '
                                              'incl. split-K finish)' if args.dtype == 'bf16x3' else 'gemm_kernel / gemm_multi_kernel<%s> (gast_gemm, gast_gemm_multi; incl. split-K finish)') % args.dtype, 'bound': bound, 'achieved': round(ach, 2),
                                   'peak': peak, 'unit': unit, 'frac': round(ach / peak, 4), 'traffic': traffic, 'traffic_source': tsrc,
                                   'launches_per_step': gm['launches'] / tsteps, 'avg_launch_us': round(avg_ms * 1e3, 2),
                                   'avg_launch_us_method': ('hipGraph replay of the step\'s gast_gemm launches alone, one HIP-event pair around 20 replays' if gemm_replay_ms else 'eager HIP-event pairs'),
                                   'eager_event_pair_avg_launch_us': round(eager_avg_ms * 1e3, 2), 'event_pair_overhead_us_subtracted': round(timer.overhead_ms * 1e3, 2),
                                   'alg_gflop_per_launch': round(gm['flops'] / gm['launches'] / 1e9, 3),
                                   'alg_mb_per_launch': round(gm['bytes'] / gm['launches'] / 1e6, 3)}
            out['kernels_note'] = ('per-op durations from an EAGER pass with a HIP-event pair around every launch (events cannot be recorded inside a '
                                   'replayed graph): the GPU clocks down between eager launches, so the column sums to more than ms_per_step; '
                                   'the replayed step itself is broken down in profiles/r06_*_step_summary.txt / _timeline.txt (rocprofv3 --kernel-trace)')
            out['kernels'] = {k: {'launches_per_step': v['launches'] / tsteps, 'ms_per_step': round(v['ms'] / tsteps, 4),
                                  'roofline_ms_per_step': round(v['roof_ms'] / tsteps, 4)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])}
            out['kernels_ms_per_step_total'] = round(sum(v['ms'] for v in agg.values()) / tsteps, 4)
            tot_roof = sum(v['roof_ms'] for v in agg.values()) / tsteps
            out['path_roofline'] = {'sum_kernel_roofline_ms_per_step': round(tot_roof, 4), 'frac_of_step': round(tot_roof / ms, 4)}
        if fwd_ms and args.config == 'cfg1' and C == 128:
            # 793.4 MB compulsory bf16 traffic (1587 MB fp32) and 160.7 GFLOP per B=128 forward of the dilated model
            fb = (793.4e6 if args.dtype in ('bf16', 'fp8') else 1586.8e6) * (B / 128.0)     # (bf16x3 stores fp32)
            ff = 160.7e9 * (B / 128.0)
            roof = max(fb / (HBM_PEAK_GBS * 1e9), ff / (MFMA_PEAK_TFLOPS[args.dtype] * 1e12)) * 1e3
            out['forward_only'] = {'ms': round(fwd_ms, 4), 'sequences_per_s': round(B / fwd_ms * 1e3, 1), 'roofline_ms': round(roof, 4),
                                   'frac_of_roofline': round(roof / fwd_ms, 4),
                                   'note': 'train-mode forward (batch-stat BN, dropout) of the same model and batch, own hipGraph; '
                                           'roofline = max(HBM, MFMA) of the algorithmic work of SURVEY.md App. C'
                                           + ('' if args.variant == 'dilated' else ' (dilated-model figure; the strided twin does 0.31x the work)')}
        if cpu is not None:
            out['cpu_baseline'] = cpu
        if world == 1 and not dry and not args.no_stock_baseline and args.config == 'cfg1':
            try:
                del model, opt, sync, graphs
                torch.cuda.empty_cache()
                out['stock_pytorch_rocm'] = stock_gpu_baseline(full=args.stock_baseline)
                sp_ = out['stock_pytorch_rocm']['fp32']['sequences_per_s']
                out['stock_pytorch_rocm']['this_path_over_stock_fp32'] = round(value / sp_, 1) if sp_ else None
                sa_ = out['stock_pytorch_rocm'].get('autocast_bf16', {}).get('sequences_per_s')
                out['stock_pytorch_rocm']['this_path_over_stock_autocast_bf16'] = round(value / sa_, 1) if sa_ else None
            except Exception as e:   # noqa: BLE001 -- never lose the bench line to this leg
                out['stock_pytorch_rocm'] = {'error': str(e).splitlines()[0][:200]}
        if collective:
            # RCCL prints its version banner through C stdio, which would otherwise be flushed at exit, AFTER the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if collective:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
