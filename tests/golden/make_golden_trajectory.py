#!/usr/bin/env python3
"""P6 of SURVEY.md section 8c: an N-step training trajectory of the REFERENCE (model + common.loss.mpjpe + optim.Adam(amsgrad=True),
reference main.py:227-239, trainval.py:78) on fixed batches, dropout 0.  Records the initial state, the batches, the loss of every
step and the final eval-mode prediction.  Build container only: python tests/golden/make_golden_trajectory.py"""
import os
import numpy as np
import torch
from make_golden import import_reference, perturb, PARENTS, HERE

gast_net, Skeleton, adj_mx_from_skeleton, mpjpe = import_reference()
torch.set_num_threads(4)
J, arc, ch, B, T, steps = 17, (3, 3, 3), 16, 32, 27, 12
torch.manual_seed(77)
gen = torch.Generator().manual_seed(99)
adj = adj_mx_from_skeleton(Skeleton(parents=list(PARENTS[J]), joints_left=[], joints_right=[]))
model = gast_net.SpatioTemporalModelOptimized1f(adj, J, 2, J, filter_widths=list(arc), causal=False, dropout=0.0, channels=ch)
perturb(model, gen)
out = {'state/' + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
xs = [torch.rand(B, T, J, 2, generator=gen) * 2 - 1 for _ in range(3)]
ys = []
for _ in range(3):
    y = torch.randn(B, 1, J, 3, generator=gen) * 0.3
    y[:, :, 0] = 0
    ys.append(y)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=True)
model.train()
losses = []
for s in range(steps):
    opt.zero_grad()
    loss = mpjpe(model(xs[s % 3]), ys[s % 3])
    loss.backward()
    opt.step()
    losses.append(loss.item())
model.eval()
with torch.no_grad():
    y_final = model(xs[0])
out.update(x=np.stack([x.numpy() for x in xs]), y3d=np.stack([y.numpy() for y in ys]), losses=np.array(losses, dtype=np.float64),
           y_final=y_final.numpy(), steps=np.array(steps))
for k, v in model.state_dict().items():
    if k in ('shrink.weight', 'expand_conv.weight', 'layers_graph_conv.1.global_graph_layer.attentions.0.C_k', 'init_bn.running_mean',
             'layers_graph_conv.2.cat_bn.running_var'):
        out['final/' + k] = v.detach().numpy().copy()
np.savez_compressed(os.path.join(HERE, 'trajectory_j17_a333_c16_str.npz'), **out)
print('losses (m):', ' '.join('%.6f' % l for l in losses))
