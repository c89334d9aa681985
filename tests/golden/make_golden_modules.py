#!/usr/bin/env python3
"""Golden fixtures for the STAND-ALONE forward of the reference's graph sub-modules, produced by the reference itself.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_modules.py

For every case: the reference module (local_attention / global_attention / gast_net / sem_graph_conv, imported unmodified) is
built under a fixed seed, its parameters and BatchNorm buffers are pushed away from their trivial initial values, and we record
the state_dict, the input, the eval-mode output, the train-mode output (no dropout: batch-statistics BatchNorm), the
BatchNorm buffers after that train-mode call, and -- the modules are trainable -- the gradients of sum(y * dy) for a seeded dy with
respect to every parameter and to the input (train mode, from the recorded state).  tests/test_modules_gpu.py loads the state into the MI355X modules and compares.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import PARENTS, import_reference  # noqa: E402

# name, module kind, J, C, kwargs, input shape builder
CASES = [
    ('mod_semch_j17_c32', 'semch', 17, 32, {}),
    ('mod_semch_bias_j15_c16', 'semch', 15, 16, {'bias': True, 'cout': 24}),
    ('mod_local_j17_c32', 'local', 17, 32, {}),
    ('mod_local_j19_c16', 'local', 19, 16, {}),
    ('mod_global_head_j17_c32', 'head', 17, 32, {'inter': 8}),
    ('mod_global_head_wide_j17_c32', 'head', 17, 32, {'inter': 16}),
    ('mod_multi_global_j17_c32', 'multi', 17, 32, {}),
    ('mod_single_global_j16_c32', 'single', 16, 32, {}),
    ('mod_gab_j17_c32', 'gab', 17, 32, {}),
    ('mod_gab_j15_c64', 'gab', 15, 64, {}),
    ('mod_semgc_j17_c32', 'semgc', 17, 32, {}),
    ('mod_sem_local_j17_c32', 'semlocal', 17, 32, {}),
]


def perturb(mod, gen):
    with torch.no_grad():
        for name, p in mod.named_parameters():
            leaf = name.split('.')[-1]
            owner = name.split('.')[-2] if '.' in name else ''
            if 'bn' in owner and leaf == 'weight':
                p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
            elif 'bn' in owner and leaf == 'bias':
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            elif leaf == 'C_k':
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            elif leaf == 'e':
                p.copy_(1.0 + torch.randn(p.shape, generator=gen) * 0.3)
            elif leaf == 'bias':
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
        for name, b in mod.named_buffers():
            if name.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
            elif name.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)


def main():
    gast_net, Skeleton, adj_mx_from_skeleton, _ = import_reference()
    from model import local_attention, global_attention, sem_graph_conv
    torch.set_num_threads(4)
    for idx, (name, kind, J, C, kw) in enumerate(CASES):
        torch.manual_seed(7000 + idx)
        gen = torch.Generator().manual_seed(8000 + idx)
        adj = adj_mx_from_skeleton(Skeleton(parents=list(PARENTS[J]), joints_left=[], joints_right=[]))
        B, T = 3, 5
        if kind == 'semch':
            sym = local_attention.LocalGraph(adj, C, C).gcn_con.adj[0].clone()          # a real (J, J) pattern: the connection graph
            mod = local_attention.SemCHGraphConv(C, kw.get('cout', C), sym, bias=kw.get('bias', False))
            x = torch.randn(B, T, J, C, generator=gen)
        elif kind == 'local':
            mod = local_attention.LocalGraph(adj, C, C, None)
            x = torch.randn(B, T, J, C, generator=gen)
        elif kind == 'head':
            mod = global_attention.GlobalGraph(adj, C, kw['inter'])
            x = torch.randn(B * T, C, J, generator=gen)
        elif kind == 'multi':
            mod = global_attention.MultiGlobalGraph(adj, C, C // 4, None)
            x = torch.randn(B, T, J, C, generator=gen)
        elif kind == 'single':
            mod = global_attention.SingleGlobalGraph(adj, C, C, None)
            x = torch.randn(B, T, J, C, generator=gen)
        elif kind == 'gab':
            mod = gast_net.GraphAttentionBlock(adj, C, C, p_dropout=0.0)
            x = torch.randn(B, C, T, J, generator=gen)
        elif kind == 'semgc':
            pat = sem_graph_conv.LocalGraph(adj, C, C).gcn_con.adj.clone()
            mod = sem_graph_conv.SemGraphConv(C, C, pat)
            x = torch.randn(B, T, J, C, generator=gen)
        elif kind == 'semlocal':
            mod = sem_graph_conv.LocalGraph(adj, C, C, None)
            x = torch.randn(B, T, J, C, generator=gen)
        perturb(mod, gen)
        out = {'x': x.numpy(), 'adj': adj.numpy().astype(np.float32)}
        for k, v in mod.state_dict().items():
            out['state/' + k] = v.detach().numpy().copy()
        with torch.no_grad():
            mod.eval()
            out['y_eval'] = mod(x.clone()).numpy()
            mod.train()
            out['y_train'] = mod(x.clone()).numpy()
        for k, v in mod.state_dict().items():
            if 'running_' in k or 'num_batches' in k:
                out['post/' + k] = v.detach().numpy().copy()
        # gradients of the reference module (the sub-modules are trainable nn.Modules): train mode from the
        # recorded state, loss = sum(y * dy) with a seeded dy -> every parameter gradient and the input gradient
        mod.load_state_dict({k[len('state/'):]: torch.from_numpy(v) for k, v in out.items() if k.startswith('state/')})
        mod.train()
        xg = x.clone().requires_grad_(True)
        y = mod(xg)
        dy = torch.randn(y.shape, generator=gen)
        (y * dy).sum().backward()
        out['dy'] = dy.numpy()
        out['dx'] = xg.grad.numpy().copy()
        for k, p in mod.named_parameters():
            out['grad/' + k] = p.grad.numpy().copy()
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print('%-32s x %s -> y %s' % (name, tuple(x.shape), out['y_eval'].shape))


if __name__ == '__main__':
    main()
