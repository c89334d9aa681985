"""How much of a gemm_big launch is the TAIL (blocks not a multiple of the resident slots)?  One shape of the step (G4 of the second stage:
N = 512, K = 256 + 512, STATS epilogue; 128 x 256 tile, 2 blocks per CU = 512 slots) at row counts that give 512 / 646 / 768 / 1024 blocks,
and G1 of the first stage (N = 648, K = 128: 3 column tiles) at 510 / 1275 / 1536 blocks.  Prints us per launch and us per 512 blocks.
python scripts/gemm_big_fill.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd')):
    sys.path.insert(0, p)
import torch
from gast_hip.binding import HipOps, RowMap
ops = HipOps(); ops.x3 = True
J = 17
g = torch.Generator().manual_seed(0)


def run(tag, N, Ks, epi, tilesM):
    T = 1
    B = tilesM * 128 // J            # rows = B * J <= tilesM * 128 (the last row tile is partial)
    M = B * J
    tm = (M + 127) // 128
    segs = []
    for K in Ks:
        A = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        segs.append(dict(A=A, K=K, map=RowMap(T, 1, 0), W=ops.x3_weight(W, True)))
    C = torch.empty(M, N).cuda()
    part = torch.zeros(tm, N, 2).cuda()
    kw = dict(epi=epi, partials=part if epi else None)
    assert ops.gemm_path((B, T, J), N, segs, C, RowMap(T, 1, 0), **kw) == 1
    for _ in range(3): ops.gemm((B, T, J), N, segs, C, RowMap(T, 1, 0), **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm((B, T, J), N, segs, C, RowMap(T, 1, 0), **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nblk = tm * ((N + 255) // 256)
    print('%-6s M=%6d blocks=%5d (%.2f x 512): %6.1f us   %5.1f us per 512 blocks' % (tag, M, nblk, nblk / 512.0, us, us * 512.0 / nblk))


for tmn in (256, 323, 384, 512):
    run('g4s1', 512, [256, 512], 1, tmn)
for tmn in (170, 256, 425, 512):
    run('g1s0', 648, [128], 0, tmn)
