"""Ablation of the large-M x3 GEMM (csrc/gemm_big.hip) on one of the step's shapes: GAST_GEMM_BIG_ABLATE=<bits> python scripts/gemm_big_ablate.py [shape]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd')):
    sys.path.insert(0, p)
import torch
from gast_hip.binding import HipOps, RowMap
ops = HipOps(); ops.x3 = True
shape = sys.argv[1] if len(sys.argv) > 1 else 'g4s1'
B, T, J = 128, 19, 17
SH = {'g4s1': (512, [256, 512], 1), 'g1s1': (1288, [256], 0), 'conv': (256, [256, 256, 256], 1), 'g1s0': (648, [128], 0), 'k2048': (256, [2048], 0)}
N, Ks, epi = SH[shape]
if shape == 'g1s0': T = 25
M = B * T * J
g = torch.Generator().manual_seed(0)
segs = []
for K in Ks:
    A = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    segs.append(dict(A=A, K=K, map=RowMap(T, 1, 0), W=ops.x3_weight(W)))
C = torch.empty(M, N).cuda()
part = torch.zeros((M + 127) // 128, N, 2).cuda()
kw = dict(epi=epi, partials=part if epi else None)
if os.environ.get('GAST_PRINT_ADDR'):      # (fault hunting: where every operand lives)
    for i, sg in enumerate(segs):
        print('seg %d: A %#x..%#x  W %#x..%#x  image %#x..%#x' % (i, sg['A'].data_ptr(), sg['A'].data_ptr() + sg['A'].numel() * 4, sg['W'].t.data_ptr(), sg['W'].t.data_ptr() + sg['W'].t.numel() * 4, sg['W'].img.data_ptr(), sg['W'].img.data_ptr() + sg['W'].img.numel() * 2), flush=True)
    print('C %#x..%#x  partials %#x..%#x' % (C.data_ptr(), C.data_ptr() + C.numel() * 4, part.data_ptr(), part.data_ptr() + part.numel() * 4), flush=True)
assert ops.gemm_path((B, T, J), N, segs, C, RowMap(T, 1, 0), **kw) == 1
for _ in range(3): ops.gemm((B, T, J), N, segs, C, RowMap(T, 1, 0), **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.gemm((B, T, J), N, segs, C, RowMap(T, 1, 0), **kw)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
fl = 2.0 * M * N * sum(Ks)
print('%s ablate=%s  M=%d N=%d K=%s: %.1f us  %.1f TF/s (x3: %.0f TF-eq/s)' % (shape, os.environ.get('GAST_GEMM_BIG_ABLATE', '0'), M, N, Ks, us, fl / us / 1e6, 3 * fl / us / 1e6))
