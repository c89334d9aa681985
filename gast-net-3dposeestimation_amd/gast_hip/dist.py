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
    """Makes every `p.grad` a view into one flat buffer and averages that buffer over the process group."""

    def __init__(self, params, process_group=None, model=None):
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
        if model is not None and hasattr(model, '_runner'):
            if [id(p) for p in model.parameters()] != [id(p) for p in self.params]:
                raise ValueError('FlatGradAllReduce(model=...) needs params == list(model.parameters())')
            model._runner.grad_sink = self.flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.force = False          # True: issue the collective even for a single rank (bench.py --force-collective)

    def zero_(self):
        self.flat.zero_()

    def sync(self):
        """sum over ranks / world size, in place; no-op for a single process."""
        if self.world > 1 or self.force:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)

    def nbytes(self):
        return self.flat.numel() * 4


def shard_batch(n_items, rank, world):
    """Indices of the global batch owned by `rank` (reference ChunkedGenerator pairs are independent units)."""
    return list(range(rank, n_items, world))
