"""Time (and check against torch fp32 matmuls) gast_wgrad_multi on the weight-gradient job sets of the B=128 step: one launch per
stage.  The tile edge / block budget are read from the environment by the library (GAST_WGRAD_TILE, GAST_WGRAD_BLOCKS[256]), so
run one process per setting:   GAST_WGRAD_TILE=256 python scripts/wgrad_multi_bench.py [s0|s1|s2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd')):
    sys.path.insert(0, p)
import torch
from gast_hip.binding import HipOps, RowMap, Dropout, dropout_params
ops = HipOps()
B, J = 128, 17
dt = torch.bfloat16
if os.environ.get('GAST_HIP_DTYPE') == 'bf16x3':      # fp32 storage, split-bf16 products
    dt = torch.float32
    ops.x3 = True
# stage -> (Tn, [(R, [(S, T_total_of_Q, t_off)])])
SETS = {
    's0': (25, [(256, [(128, 25, 0), (256, 25, 0)]), (128, [(256, 25, 0)]), (128, [(128, 25, 0)]), (648, [(128, 25, 0)])]),
    's1': (19, [(512, [(256, 19, 0), (512, 19, 0)]), (256, [(512, 19, 0)]), (256, [(256, 19, 0)]), (1288, [(256, 19, 0)]),
                (256, [(256, 19, 0)]), (256, [(256, 25, 0), (256, 25, 3), (256, 25, 6)])]),
    's2': (1, [(8, [(1024, 1, 0)]), (1024, [(512, 1, 0), (1024, 1, 0)]), (512, [(1024, 1, 0)]), (512, [(512, 1, 0)]),
               (2568, [(512, 1, 0)]), (512, [(512, 1, 0)]), (512, [(512, 19, 0), (512, 19, 9), (512, 19, 18)])]),
}
th, ik = dropout_params(0.05)
seed = torch.tensor([5], dtype=torch.int32).cuda()
reps = int(os.environ.get('GAST_MB_REPS', 20))
for name in (sys.argv[1:] or ['s0', 's1', 's2']):
    Tn, jobs_spec = SETS[name]
    M = B * Tn * J
    g = torch.Generator(device='cuda').manual_seed(1)
    jobs, refs, flops = [], [], 0.0
    for R, segs in jobs_spec:
        P = (torch.randn(M, R, device='cuda', generator=g) * 0.5).to(dt)
        sg, col = [], 0
        srcs = {}
        for S, Tt, toff in segs:
            key = (S, Tt)
            if key not in srcs:
                srcs[key] = (torch.randn(B * Tt * J, S, device='cuda', generator=g) * 0.5).to(dt)
            sg.append(dict(Q=srcs[key], S=S, map=RowMap(Tt, 1, toff), pro=0, wcol0=col))
            col += S
        dW = torch.zeros(R, col, device='cuda')
        jobs.append(dict(dom=(B, Tn, J), P=P, R=R, pmap=RowMap(Tn, 1, 0), segs=sg, dW=dW, drop=Dropout(seed, th, ik), zero_first=False))
        flops += 2.0 * M * R * col
        # reference
        cols = []
        for (S, Tt, toff), s_ in zip(segs, sg):
            q = s_['Q'].view(B, Tt, J, S)[:, toff:toff + Tn].reshape(M, S)
            cols.append(P.float().t() @ q.float())
        refs.append(torch.cat(cols, 1))
    ops.wgrad_multi(jobs)
    torch.cuda.synchronize()
    worst = 0.0
    for j, r in zip(jobs, refs):
        err = (j['dW'] - r).abs().max().item() / r.abs().max().item()
        worst = max(worst, err)
    for _ in range(2):
        ops.wgrad_multi(jobs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.wgrad_multi(jobs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print('%s  tile=%s blocks256=%s  M=%6d  %7.1f us  %7.1f TF/s   max rel err %.2e' % (
        name, os.environ.get('GAST_WGRAD_TILE', 'auto'), os.environ.get('GAST_WGRAD_BLOCKS256', '-'), M, us, flops / us / 1e6, worst), flush=True)
