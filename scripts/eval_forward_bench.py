import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gast-net-3dposeestimation_amd')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
os.environ['GAST_HIP_DTYPE'] = 'bf16'
from tests_helpers import PARENTS
from model.gast_net import SpatioTemporalModel
from bench import adj_from_parents      # (the product's callers build adj themselves; nothing under oracle/ is used here)
m = SpatioTemporalModel(adj_from_parents(PARENTS[17]), 17, 2, 17, filter_widths=[3,3,3], channels=128, dropout=0.05).cuda().eval()
for shape in [(128, 27, 17, 2), (2, 2026, 17, 2)]:
    x = (torch.rand(*shape) * 2 - 1).cuda()
    with torch.no_grad():
        for _ in range(3): m(x)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            y = m(x)
        for _ in range(3): g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
    print(shape, 'eval forward %.3f ms' % (e0.elapsed_time(e1) / 20), tuple(y.shape))
