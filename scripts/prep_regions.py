"""Sizes of the regions the pass prologues (gast_prep) zero-fill in one training step (GPU box): how the 84.7 MB zero fill of the last
dilated level's input gradient was found (round 6, gast_bn_bwd_apply_frames).  Usage: python scripts/prep_regions.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd')):
    sys.path.insert(0, p)
import torch
os.environ['GAST_HIP_DTYPE'] = 'bf16x3'
os.environ['GAST_HIP_GRAPH'] = '0'
from bench import adj_from_parents, PARENTS17
from model.gast_net import SpatioTemporalModel
torch.manual_seed(0)
m = SpatioTemporalModel(adj_from_parents(PARENTS17), 17, 2, 17, filter_widths=[3, 3, 3], channels=128, dropout=0.05).cuda().train()
x = (torch.rand(128, 27, 17, 2) * 2 - 1).cuda()
y = torch.randn(128, 1, 17, 3).cuda()
ops = m._runner.engine.ops
orig = ops.prep
def prep(zero, seed=None, pad=None):
    print('prep:', [(tuple(t.shape), t.numel() * t.element_size() / 1e6) for t in zero], 'seed' if seed is not None else '', 'pad' if pad is not None else '')
    return orig(zero, seed=seed, pad=pad)
ops.prep = prep
from gast_hip.loss import mpjpe
for _ in range(2):
    m.zero_grad(); mpjpe(m(x), y).backward()
torch.cuda.synchronize()
