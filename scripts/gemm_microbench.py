"""Time gast_gemm / gast_wgrad on the real shapes of the B=128 forward/backward (HIP events, N reps each)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd')):
    sys.path.insert(0, p)
import torch
from gast_hip.binding import HipOps, RowMap
ops = HipOps()
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
reps = 20
B, J = 128, 17
def ident(T): return RowMap(T, 1, 0)
# (name, Tn, N, [(K, T_total, t_stride, t_off, pro)], epi)
SHAPES = [
 ('G1 s0  K128 N648', 25, 648, [(128, 25, 1, 0, 0)], 0),
 ('G2 s0  K256 N128 pro', 25, 128, [(256, 25, 1, 0, 1)], 1),
 ('G3 s0  K128 N128', 25, 128, [(128, 25, 1, 0, 0)], 1),
 ('G4 s0  K384 N256 3seg', 25, 256, [(128, 25, 1, 0, 0), (128, 25, 1, 0, 2), (128, 25, 1, 0, 2)], 1),
 ('conv1  K768 N256 taps', 19, 256, [(256, 25, 1, 0, 1), (256, 25, 1, 3, 1), (256, 25, 1, 6, 1)], 1),
 ('G1 s1  K256 N1288', 19, 1288, [(256, 19, 1, 0, 0)], 0),
 ('G4 s1  K768 N512 3seg', 19, 512, [(256, 19, 1, 0, 0), (256, 19, 1, 0, 2), (256, 19, 1, 0, 2)], 1),
 ('conv2  K1536 N512 M2176', 1, 512, [(512, 19, 1, 0, 1), (512, 19, 1, 9, 1), (512, 19, 1, 18, 1)], 1),
 ('G4 s2  K1536 N1024 M2176', 1, 1024, [(512, 1, 1, 0, 0), (512, 1, 1, 0, 2), (512, 1, 1, 0, 2)], 1),
 ('dG4 s1 bwd K512 N256', 19, 256, [(512, 19, 1, 0, 0)], 2),
]
from gast_hip.binding import Dropout, dropout_params
th, ik = dropout_params(0.05)
seed = torch.tensor([5], dtype=torch.int32).cuda()
out = []
ONLY = os.environ.get('GAST_MB_ONLY')
reps = int(os.environ.get('GAST_MB_REPS', reps))
for name, Tn, N, segs, epi in ([] if (len(sys.argv) > 2 and sys.argv[2] == 'wgrad') else [s_ for s_ in SHAPES if not ONLY or ONLY in s_[0]]):
    M = B * Tn * J
    sg = []
    Ktot = 0
    abytes = 0
    seen = set()
    for si, (K, Tt, ts, toff, pro) in enumerate(segs):
        A = torch.randn(B * Tt * J, K, device='cuda').to(dt)
        W = (torch.randn(N, K, device='cuda') / K ** 0.5).to(dt)
        sg.append(dict(A=A, K=K, map=RowMap(Tt, ts, toff), W=W, pro=pro, scale=torch.rand(K, device='cuda') + 0.5, shift=torch.randn(K, device='cuda') * 0.1, salt=si))
        Ktot += K
    if len(segs) == 3 and segs[0][1] != Tn:   # conv taps share one tensor
        for s_ in sg[1:]:
            s_['A'] = sg[0]['A']
        abytes = B * segs[0][1] * J * segs[0][0] * A.element_size()
    else:
        abytes = sum(M * s_['K'] for s_ in sg) * A.element_size()
    C = torch.empty(M, N, device='cuda', dtype=dt)
    part = torch.empty(ops.gemm_row_blocks(M), N, 2, device='cuda')
    X = torch.randn(M, N, device='cuda').to(dt) if epi == 2 else None
    kw = dict(epi=epi, partials=part if epi else None, drop=Dropout(seed, th, ik))
    if epi == 2:
        kw.update(X=X, xscale=torch.ones(N, device='cuda'), xshift=torch.zeros(N, device='cuda'))
    for _ in range(3):
        ops.gemm((B, Tn, J), N, sg, C, ident(Tn), **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm((B, Tn, J), N, sg, C, ident(Tn), **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * M * N * Ktot
    by = abytes + (N * Ktot + M * N) * C.element_size() + (M * N * C.element_size() if epi == 2 else 0)
    print('%-28s M=%6d  %7.1f us  %7.1f TF  %6.2f TB/s (alg)  blocks=%d' % (name, M, us, fl / us / 1e6, by / us / 1e6, ((M + 127) // 128) * ((N + 127) // 128)), flush=True)

# ---- weight gradients: (name, Tn, R, [(S, T_total, t_stride, t_off, pro)])
WSHAPES = [
 ('wG4 s1 R512 S768 3seg', 19, 512, [(256, 19, 1, 0, 0), (256, 19, 1, 0, 2), (256, 19, 1, 0, 2)]),
 ('wG1 s1 R1288 S256', 19, 1288, [(256, 19, 1, 0, 0)]),
 ('wconv1 R256 S768 taps', 19, 256, [(256, 25, 1, 0, 1), (256, 25, 1, 3, 1), (256, 25, 1, 6, 1)]),
 ('wG2 s1 R256 S512 pro', 19, 256, [(512, 19, 1, 0, 1)]),
 ('wG3 s1 R256 S256', 19, 256, [(256, 19, 1, 0, 0)]),
 ('wG4 s0 R256 S384 3seg', 25, 256, [(128, 25, 1, 0, 0), (128, 25, 1, 0, 2), (128, 25, 1, 0, 2)]),
 ('wG1 s0 R648 S128', 25, 648, [(128, 25, 1, 0, 0)]),
 ('wG2 s0 R128 S256 pro', 25, 128, [(256, 25, 1, 0, 1)]),
]
if len(sys.argv) > 2 and sys.argv[2] == 'wgrad':
    for name, Tn, R, segs in WSHAPES:
        M = B * Tn * J
        P = torch.randn(M, R, device='cuda').to(dt)
        sg, Stot, col = [], 0, 0
        for si, (S, Tt, ts, toff, pro) in enumerate(segs):
            Q = torch.randn(B * Tt * J, S, device='cuda').to(dt)
            sg.append(dict(Q=Q, S=S, map=RowMap(Tt, ts, toff), pro=pro, scale=torch.rand(S, device='cuda') + 0.5,
                           shift=torch.randn(S, device='cuda') * 0.1, salt=si, wcol0=col))
            col += S
        if len(segs) == 3 and segs[0][1] != Tn:
            for s_ in sg[1:]:
                s_['Q'] = sg[0]['Q']
        dW = torch.zeros(R, col, device='cuda')
        for _ in range(3):
            ops.wgrad((B, Tn, J), P, R, ident(Tn), sg, dW, drop=Dropout(seed, th, ik), zero_first=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.wgrad((B, Tn, J), P, R, ident(Tn), sg, dW, drop=Dropout(seed, th, ik), zero_first=False)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        fl = 2.0 * M * R * col
        by = (M * R + M * col) * P.element_size()
        print('%-28s M=%6d  %7.1f us  %7.1f TF  %6.2f TB/s (alg)' % (name, M, us, fl / us / 1e6, by / us / 1e6), flush=True)
