#!/usr/bin/env python3
"""Generate golden input/output/gradient fixtures by running the REFERENCE itself.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference model (`/root/reference/model/gast_net.py`) is imported unmodified, with a stub for its
unused `torchsummary` import (gast_net.py:2).  For each small configuration we record

* the configuration (skeleton parents, filter widths, channels, causal flag, variant),
* the full `state_dict` (after perturbing BN affine/running stats, `C_k`, `e` and the attention biases away
  from their trivial initial values so that every parameter matters),
* the input `x` (B,T,J,2) and the target `y3d`,
* `y_eval`: `model.eval()` forward,
* `y_train`: `model.train()` forward with dropout 0 (batch-statistics BatchNorm),
* `loss`: `mpjpe(y_train, y3d)` (common/loss.py:5-11, the loss `main.py:231` trains with),
* `grad/<name>`: d loss / d parameter for every parameter (reference autograd),
* `post/<name>`: BatchNorm buffers after the train-mode forward (momentum update check).

Nothing here is imported by the product; the fixtures are plain `.npz` files read by tests/.
"""
import os
import sys
import types
import json

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

PARENTS = {
    17: [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15],            # reconstruction.py:95
    19: [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 10, 13, 14, 10, 16, 17],  # reconstruction.py:87
    15: [-1, 0, 1, 2, 3, 1, 5, 6, 0, 8, 9, 0, 11, 12, 1],                    # common/humaneva_dataset.py:7
    16: [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 8, 10, 11, 8, 13, 14],               # h36m_dataset.py:267-277 (SH, nose removed)
}

# name, J, arc, channels, causal, variant, B, T
CONFIGS = [
    ('j17_a333_c16_dil', 17, (3, 3, 3), 16, False, 'dilated', 3, 29),
    ('j17_a333_c16_str', 17, (3, 3, 3), 16, False, 'strided', 5, 27),
    ('j17_a333_c16_dil_causal', 17, (3, 3, 3), 16, True, 'dilated', 2, 30),
    ('j17_a333_c16_str_causal', 17, (3, 3, 3), 16, True, 'strided', 4, 27),
    ('j19_a33_c32_dil', 19, (3, 3), 32, False, 'dilated', 3, 11),
    ('j15_a333_c16_dil', 15, (3, 3, 3), 16, False, 'dilated', 2, 27),
    ('j16_a33_c16_str', 16, (3, 3), 16, False, 'strided', 4, 9),
    ('j17_a53_c16_dil', 17, (5, 3), 16, False, 'dilated', 2, 17),
    ('j17_a3333_c8_dil', 17, (3, 3, 3, 3), 8, False, 'dilated', 2, 83),
    # variant 'dense' = SpatioTemporalModel(dense=True), the ablation of gast_net.py:145-146 (temporal kernels 7 and 19 wide)
    ('j17_a333_c16_dense', 17, (3, 3, 3), 16, False, 'dense', 3, 29),
    ('j19_a33_c32_dense_causal', 19, (3, 3), 32, True, 'dense', 2, 12),
    # the five-level plan of the shipped 243-frame model (reconstruction.py:225-227: arc 3,3,3,3,3), small width, T = RF + 2
    ('j17_a33333_c8_dil', 17, (3, 3, 3, 3, 3), 8, False, 'dilated', 2, 245),
]


def import_reference():
    stub = types.ModuleType('torchsummary')
    stub.summary = lambda *a, **k: None
    sys.modules['torchsummary'] = stub
    sys.path.insert(0, REF)
    from model import gast_net  # noqa
    from common.skeleton import Skeleton
    from common.graph_utils import adj_mx_from_skeleton
    from common.loss import mpjpe
    return gast_net, Skeleton, adj_mx_from_skeleton, mpjpe


def perturb(model, gen):
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('_bn.weight') or '.bn_1.weight' in name or '.bn_2.weight' in name or \
                    name.startswith('layers_bn.') and name.endswith('weight') or name in ('init_bn.weight', 'expand_bn.weight'):
                p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
            elif name.endswith('bn.bias') or '.bn_1.bias' in name or '.bn_2.bias' in name or \
                    name.startswith('layers_bn.') and name.endswith('bias') or name in ('init_bn.bias', 'expand_bn.bias'):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            elif name.endswith('C_k'):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
            elif name.endswith('.e'):
                p.copy_(1.0 + torch.randn(p.shape, generator=gen) * 0.3)
            elif name.endswith('.bias'):  # g/theta/phi conv1d biases
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
        for name, b in model.named_buffers():
            if name.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
            elif name.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)


def main():
    gast_net, Skeleton, adj_mx_from_skeleton, mpjpe = import_reference()
    torch.set_num_threads(4)
    only = set(sys.argv[1:])          # optional: regenerate only these fixtures (the others keep their files and index entries)
    old_index = json.load(open(os.path.join(HERE, 'index.json'))) if only else {}
    index = {}
    for (name, J, arc, ch, causal, variant, B, T) in CONFIGS:
        torch.manual_seed(1000 + len(index))
        gen = torch.Generator().manual_seed(4321 + len(index))
        skel = Skeleton(parents=list(PARENTS[J]), joints_left=[], joints_right=[])
        adj = adj_mx_from_skeleton(skel)
        if only and name not in only:
            index[name] = old_index[name]          # keeps the position-dependent seeds of the later entries
            continue
        if variant == 'strided':
            model = gast_net.SpatioTemporalModelOptimized1f(adj, J, 2, J, filter_widths=list(arc), causal=causal, dropout=0.0, channels=ch)
        else:
            model = gast_net.SpatioTemporalModel(adj, J, 2, J, filter_widths=list(arc), causal=causal, dropout=0.0, channels=ch,
                                                 dense=(variant == 'dense'))
        perturb(model, gen)
        x = torch.rand(B, T, J, 2, generator=gen) * 2 - 1

        out = {'x': x.numpy().copy()}
        for k, v in model.state_dict().items():
            out['state/' + k] = v.detach().numpy().copy()

        model.eval()
        with torch.no_grad():
            y_eval = model(x)
        out['y_eval'] = y_eval.numpy().copy()

        model.train()
        y_train = model(x)
        y3d = torch.randn(y_train.shape, generator=gen) * 0.3
        y3d[:, :, 0] = 0  # main.py:225
        loss = mpjpe(y_train, y3d)
        loss.backward()
        out['y3d'] = y3d.numpy().copy()
        out['y_train'] = y_train.detach().numpy().copy()
        out['loss'] = np.float64(loss.item())
        for k, p in model.named_parameters():
            out['grad/' + k] = p.grad.detach().numpy().copy()
        for k, b in model.named_buffers():
            out['post/' + k] = b.detach().numpy().copy()
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        nparam = sum(p.numel() for p in model.parameters())
        index[name] = dict(J=J, parents=PARENTS[J], arc=list(arc), channels=ch, causal=causal, variant=variant,
                           B=B, T=T, T_out=int(y_train.shape[1]), receptive_field=int(model.receptive_field()),
                           n_params=int(nparam), n_state=len(model.state_dict()), loss=float(loss.item()))
        print(name, 'params', nparam, 'out', tuple(y_train.shape), 'loss %.6f' % loss.item())

    # known-answer numbers the reference prints/relies on: parameter counts of the shipped shapes
    counts = {}
    for J in (() if only else (17, 19, 15)):
        skel = Skeleton(parents=list(PARENTS[J]), joints_left=[], joints_right=[])
        adj = adj_mx_from_skeleton(skel)
        m = gast_net.SpatioTemporalModel(adj, J, 2, J, filter_widths=[3, 3, 3], channels=128)
        counts['J%d_a333_c128' % J] = int(sum(p.numel() for p in m.parameters()))
        if J == 17:
            keys = {k: list(v.shape) for k, v in m.state_dict().items()}
            with open(os.path.join(HERE, 'state_dict_contract_j17_a333_c128.json'), 'w') as f:
                json.dump(keys, f, indent=0)
            # adjacency known answer
            np.save(os.path.join(HERE, 'adj_j17.npy'), adj.numpy())
    index['_param_counts'] = old_index.get('_param_counts', counts) if only else counts
    with open(os.path.join(HERE, 'index.json'), 'w') as f:
        json.dump(index, f, indent=1)
    print(counts)


if __name__ == '__main__':
    main()
