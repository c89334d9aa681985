"""Accuracy of the three fp32-storage GEMM arithmetics against float64 on one forward-shaped product (A post-ReLU O(1), W ~ 1/sqrt(K)):
fp32 MFMA, bf16 hi/lo pairs (GAST_F32X3), fp16 hi/lo pairs (GAST_F32X3H).  Prints max / rms / mean signed error relative to the rms
of the result.  python scripts/gemm_pair_accuracy.py [M] [K] [N]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gast-net-3dposeestimation_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gast_hip.binding import HipOps          # noqa: E402
from gast_hip.engine import ident            # noqa: E402
from gast_hip.packer import X3Weight         # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 27 * 17
K = int(sys.argv[2]) if len(sys.argv) > 2 else 640
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ops = HipOps()
g = torch.Generator().manual_seed(1)
A = torch.relu(torch.randn(M, K, generator=g)).cuda()
W = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
ref = A.double() @ W.double().t()
rms = float(ref.pow(2).mean().sqrt())
dom = (M // 17, 1, 17)


def run(tag, x3, wt):
    ops.x3 = x3
    Cd = torch.empty(M, N, device='cuda')
    ops.gemm(dom, N, [dict(A=A, K=K, map=ident(1), W=wt)], Cd, ident(1))
    torch.cuda.synchronize()
    e = (Cd.double() - ref) / rms
    print('%-28s max %.2e  rms %.2e  mean %+.2e' % (tag, float(e.abs().max()), float(e.pow(2).mean().sqrt()), float(e.mean())))


run('fp32 MFMA', False, W)
run('bf16 pairs (gemm.hip)', True, W)
run('bf16 pairs (gemm_big)', True, ops.x3_weight(W))
run('fp16 pairs (gemm.hip)', True, X3Weight(W, None, True))
run('fp16 pairs (gemm_big)', True, ops.x3_weight(W, True))
e = (A @ W.t()).double() - ref
print('%-28s max %.2e  rms %.2e  mean %+.2e' % ('torch fp32 matmul', float(e.abs().max()) / rms, float(e.pow(2).mean().sqrt()) / rms, float(e.mean()) / rms))
