"""Batch-sharded data parallelism for the hot path: one process per GPU, replicated weights, ONE all-reduce of a flat
fp32 gradient buffer per step (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference's only parallelism is `nn.DataParallel(device_ids=[0,1])` (reference trainval.py:56-61): per-replica
BatchNorm statistics, gradients summed on device 0.  Here each rank keeps its own BatchNorm statistics as well
(reference-faithful DP) and the gradient of every parameter lives in a single contiguous buffer, so the exchange step is
a single 27.7 MB (6.9 M fp32, J=17 / arc 3,3,3 / C=128) all-reduce instead of 165 small ones.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Makes every `p.grad` a view into one flat buffer and averages that buffer over the process group.

    buckets = 1 (default): ONE all-reduce of the whole buffer in `sync()`, after backward.
    buckets > 1 (needs `model=`): the buffer is cut, in parameter order, into the ranges whose gradients are complete at the same
    point of the backward pass -- [layers_graph_conv.L-1], ..., [layers_graph_conv.1], [everything else] -- and each range is
    all-reduced on a communication stream as soon as the engine reports its stage done (gast_hip/engine.py: `stage_done`), so the
    exchange of the deepest block (2/3 of the parameters) overlaps the backward pass of the shallower stages; `sync()` then only
    makes the compute stream wait for the communication stream.  Both modes produce identical sums (tests/test_dist_cpu.py).

    The 1/world factor: applied in `sync()` by default; `attach(optimizer)` moves it into FlatAdam's `grad_scale` (one launch less
    and one pass over the buffer less per step)."""

    def __init__(self, params, process_group=None, model=None, buckets=1):
        """model: optional GAST model whose backward should accumulate straight into the flat buffer (skips 165 per-parameter
        AccumulateGrad kernels); `params` must then be `model.parameters()` in order."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.force = False          # True: issue the collective even for a single rank (bench.py --force-collective)
        self.scale_in_optimizer = False
        self.ranges = [(0, n)]      # completion order
        self._comm = None
        self._issued = 0
        self._runner = model._runner if model is not None and hasattr(model, '_runner') else None
        if model is not None and hasattr(model, '_runner'):
            if [id(p) for p in model.parameters()] != [id(p) for p in self.params]:
                raise ValueError('FlatGradAllReduce(model=...) needs params == list(model.parameters())')
            model._runner.grad_sink = self.flat
            if buckets > 1:
                self.ranges = self._stage_ranges(model, n)
                model._runner.grad_sync = self
        elif buckets > 1:
            raise ValueError('bucketed exchange needs model= (the engine reports when a stage\'s gradients are complete)')

    @staticmethod
    def _stage_ranges(model, n):
        """[(start, end)] of the flat buffer in completion order: the blocks layers_graph_conv.{L-1 .. 1} (contiguous and last in
        registration order, reference gast_net.py:155-157), then the rest [0, start of block 1)."""
        starts, off = {}, 0
        for name, p in model.named_parameters():
            if name.startswith('layers_graph_conv.'):
                starts.setdefault(int(name.split('.')[1]), off)
            off += p.numel()
        L = len(starts)
        if L < 2:
            return [(0, n)]
        ends = {s: (starts[s + 1] if s + 1 < L else n) for s in range(L)}
        assert all(starts[s] < ends[s] for s in range(L)) and ends[L - 1] == n, 'graph-conv blocks are not the tail of parameters()'
        return [(starts[s], ends[s]) for s in range(L - 1, 0, -1)] + [(0, starts[1])]

    def bucket_of_stage(self, s):
        """index into `ranges` of the bucket that is complete when stage s >= 1 of the backward pass is done"""
        return len(self.ranges) - 1 - s

    def attach(self, optimizer):
        """Fold the 1/world averaging into the optimizer's gradient scale (gast_hip.optim.FlatAdam)."""
        if not hasattr(optimizer, 'grad_scale'):
            raise TypeError('attach() needs an optimizer with a grad_scale attribute (gast_hip.optim.FlatAdam)')
        optimizer.grad_scale = 1.0 / self.world
        self.scale_in_optimizer = True
        return self

    def zero_(self, defer=False):
        """defer=True (needs model=): the fill is not launched now but as part of the pass prologue of the model's NEXT forward
        (one launch with its other zero fills, gast_prep) -- for loops that go zero_() -> forward -> backward -> step, where nothing
        reads the gradients in between."""
        if defer and self._runner is not None:
            if not any(t is self.flat for t in self._runner.pending_zero):
                self._runner.pending_zero.append(self.flat)
        else:
            self.flat.zero_()
        self._issued = 0

    def _active(self):
        return self.world > 1 or self.force

    def _no_pending_zero(self, where):
        """a zero fill parked by zero_(defer=True) must have been consumed (by the model's forward, or flushed by its backward) before
        the buffer is exchanged: reducing -- or later wiping -- gradients around an unexecuted fill would be silently wrong"""
        if self._runner is not None and any(t is self.flat for t in self._runner.pending_zero):
            raise RuntimeError('FlatGradAllReduce.%s: a deferred zero_() of this buffer is still pending -- zero_(defer=True) is for loops '
                               'that run zero_() -> model forward -> backward -> sync(); use zero_() (immediate) otherwise' % where)

    def bucket_ready(self, i):
        """Called by the model's backward (bucketed mode) when `ranges[i]` holds its final local gradients."""
        self._no_pending_zero('bucket_ready')
        if i != self._issued:
            # a second backward() before sync() (gradient accumulation) would all-reduce ranges that already hold reduced sums
            raise RuntimeError('bucketed all-reduce: bucket %d reported complete, bucket %d expected -- the bucketed exchange runs ONE '
                               'backward() per sync() (no gradient accumulation; use the monolithic mode for that)' % (i, self._issued))
        self._issued += 1
        if not self._active():
            return
        a, b = self.ranges[i]
        if self.flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.flat.device)
            self._comm.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self._comm):
                dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group)

    def sync(self):
        """sum over ranks / world size, in place; no-op for a single process."""
        self._no_pending_zero('sync')
        if not self._active():
            self._issued = 0
            return
        if len(self.ranges) > 1:
            if self._issued != len(self.ranges):
                raise RuntimeError('bucketed all-reduce: %d of %d buckets were issued by backward()' % (self._issued, len(self.ranges)))
            if self._comm is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self._comm)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if not self.scale_in_optimizer:
            self.flat.mul_(1.0 / self.world)
        self._issued = 0      # the next backward starts a new round whichever way the buffer gets zeroed (zero_(), FlatAdam.zero_grad(), ...)

    def nbytes(self):
        return self.flat.numel() * 4


def shard_batch(n_items, rank, world):
    """Indices of the global batch owned by `rank` (reference ChunkedGenerator pairs are independent units)."""
    return list(range(rank, n_items, world))
