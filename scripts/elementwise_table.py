"""Per-launch table of the step's non-GEMM launches (GPU box): rows x columns, algorithmic bytes, duration from a hipGraph replay of
that single launch (20 replays inside one event pair) and achieved GB/s against the 8 TB/s HBM roofline.
Usage: python scripts/elementwise_table.py [bf16x3|fp32|f16] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd')):
    sys.path.insert(0, p)
import torch
os.environ['GAST_HIP_DTYPE'] = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
from bench import adj_from_parents, PARENTS17
from model.gast_net import SpatioTemporalModel

torch.manual_seed(0)
m = SpatioTemporalModel(adj_from_parents(PARENTS17), 17, 2, 17, filter_widths=[3, 3, 3], channels=128, dropout=0.05).cuda().train()
g = torch.Generator().manual_seed(1234)
x = (torch.rand(B, 27, 17, 2, generator=g) * 2 - 1).cuda()
y3d = (torch.randn(B, 1, 17, 3, generator=g) * 0.3).cuda()
ops = m._runner.engine.ops
for _ in range(2):
    m.zero_grad(); torch.mean(torch.norm(m(x) - y3d, dim=-1)).backward()
torch.cuda.synchronize()

def nbytes(v, seen):
    if torch.is_tensor(v):
        key = (v.data_ptr(), v.numel())
        if key in seen:
            return 0
        seen.add(key)
        return v.numel() * v.element_size()
    if isinstance(v, dict):
        return sum(nbytes(u, seen) for u in v.values())
    if isinstance(v, (list, tuple)):
        return sum(nbytes(u, seen) for u in v)
    return 0

NAMES = ('bn_bwd_apply', 'bnrelu_apply', 'residual_fwd', 'semch_agg_fwd', 'semch_agg_bwd', 'attn_fwd', 'attn_bwd', 'expand_fwd', 'expand_bwd',
         'rowsum_multi', 'bn_bwd_fused_multi', 'bn_finalize_multi', 'bn_bwd_finalize_multi', 'unfold')
calls = []
for name in NAMES:
    if not hasattr(ops, name):
        continue
    orig = getattr(ops, name)
    def rec(*a, _n=name, _o=orig, **k):
        calls.append((_n, _o, a, k))
        return _o(*a, **k)
    setattr(ops, name, rec)
m.zero_grad(); torch.mean(torch.norm(m(x) - y3d, dim=-1)).backward()
torch.cuda.synchronize()

def timeit(fn, a, k, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(*a, **k)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn(*a, **k)
    for _ in range(3): gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

tot = 0.0
print('%-24s %-40s %9s %8s %8s' % ('op', 'tensors', 'MB(args)', 'us', 'GB/s'))
for name, fn, a, k in calls:
    seen = set()
    by = nbytes(a, seen) + nbytes(k, seen)      # every distinct tensor argument once ...
    if name == 'bn_bwd_apply':                  # ... except the in-place update: dz is read AND written, X read
        by = 3 * a[0].numel() * a[0].element_size()
    shapes = [tuple(v.shape) for v in list(a) + list(k.values()) if torch.is_tensor(v) and v.numel() > 4096][:3]
    us = timeit(fn, a, k)
    tot += us
    print('%-24s %-40s %9.1f %8.1f %8.0f' % (name, str(shapes)[:40], by / 1e6, us, by / us / 1e3))
print('total %.1f us   (MB(args): every tensor argument once, whole tensors even where a launch touches a column range; bn_bwd_apply: 3 x the tensor.  Each figure includes the ~5 us of a one-node graph replay)' % tot)
