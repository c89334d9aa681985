"""Per-launch table of the step's GEMM / weight-gradient launches (GPU box): shape, algorithmic FLOPs and bytes, duration from a
hipGraph replay of that single launch (20 replays inside one event pair), achieved TF/s and GB/s.
Usage: python scripts/gemm_table.py [bf16|bf16x3|fp32] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd')):
    sys.path.insert(0, p)
import torch
os.environ['GAST_HIP_DTYPE'] = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
from bench import adj_from_parents, PARENTS17, KernelTimer
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
kt = KernelTimer(ops)
calls = []
for name in ('gemm', 'gemm_multi', 'wgrad_multi'):
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

MAXM = int(os.environ.get('GEMM_TABLE_MAXM', '0'))      # only the launches with at most this many rows (0: all)
tot = 0.0
print('%-12s %-34s %9s %9s %8s %8s %8s' % ('op', 'shape', 'GFLOP', 'MB', 'us', 'TF/s', 'GB/s'))
for name, fn, a, k in calls:
    rows_ = (a[0][0] * a[0][1] * a[0][2]) if name == 'gemm' else (a[0][0]['dom'][0] * a[0][0]['dom'][1] * a[0][0]['dom'][2])
    if MAXM and rows_ > MAXM:
        continue
    if name == 'gemm':
        fl, by = kt.cost_gemm(*a, **k)
        dom, N, segs = a[0], a[1], a[2]
        shape = 'M=%d N=%d K=%s epi=%d' % (dom[0] * dom[1] * dom[2], N, '+'.join(str(s['K']) for s in segs), k.get('epi', 0))
    elif name == 'gemm_multi':
        fl, by = kt.cost_gemm_multi(*a, **k)
        shape = ' | '.join('M=%d N=%d K=%s' % (j['dom'][0] * j['dom'][1] * j['dom'][2], j['N'], '+'.join(str(s['K']) for s in j['segs'])) for j in a[0])
    else:
        fl, by = kt.cost_wgrad_multi(*a, **k)
        shape = '%d jobs M=%d' % (len(a[0]), a[0][0]['dom'][0] * a[0][0]['dom'][1] * a[0][0]['dom'][2])
    us = timeit(fn, a, k)
    tot += us
    print('%-12s %-34s %9.2f %9.1f %8.1f %8.1f %8.0f' % (name, shape[:34], fl / 1e9, by / 1e6, us, fl / us / 1e6, by / us / 1e3))
print('total %.1f us' % tot)
