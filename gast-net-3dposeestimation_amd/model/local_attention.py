"""Local graph attention: parameters, skeleton patterns (reference model/local_attention.py).

`SemCHGraphConv` / `LocalGraph` keep the reference's constructor signatures, parameter names, shapes and initialisers
(local_attention.py:15-33, 60-128) so that checkpoints load unchanged.  Inside SpatioTemporalModel the arithmetic of their
forward (local_attention.py:35-53, 130-151) runs in the fused HIP plan (gast_hip/engine.py: G1 + AGG + G2); called on their
own they run the same kernels as a small forward-only plan (gast_hip/modules.py).
"""
from __future__ import absolute_import, division

import math

import numpy as np
import torch
import torch.nn as nn

# joint groups of the supported skeletons (reference local_attention.py:66-87)
_GROUPS = {
    17: ([3, 6, 10, 13, 16], [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]),          # Human3.6M
    16: ([3, 6, 9, 12, 15], [4, 5, 6, 10, 11, 12], [1, 2, 3, 13, 14, 15]),           # Human3.6M from Stacked Hourglass
    15: ([4, 7, 10, 13], [2, 3, 4, 8, 9, 10], [5, 6, 7, 11, 12, 13]),                # HumanEva
    19: ([3, 4, 7, 8, 12, 15, 18], [5, 6, 7, 8, 13, 14, 15], [1, 2, 3, 4, 16, 17, 18]),  # Human3.6M + toes
}


def skeleton_patterns(adj):
    """(adj_sym, adj_con) float tensors like the reference builds at local_attention.py:92-114; only their `> 0`
    pattern is used.  `adj` is not modified."""
    J = adj.shape[0]
    if J not in _GROUPS:
        raise KeyError("The dimension of adj matrix is wrong!")
    distal, left, right = _GROUPS[J]
    a = adj.detach().to('cpu', torch.float32).clone()
    sym = torch.eye(J)
    for l, r in zip(left, right):
        sym[l, r] = 1.0
        sym[r, l] = 1.0
    is_distal = torch.zeros(J, dtype=torch.bool)
    is_distal[distal] = True
    first = a.clone()
    first[is_distal] = 0
    second = a @ a
    second[~is_distal] = 0
    return sym.to(adj.dtype), (first + second).to(adj.dtype)


def pattern_table(pat):
    """int32 table consumed by the HIP kernels (layout documented in include/gast_hip.h): CSR + CSC views of the
    0/1 pattern, edges enumerated row-major exactly like the reference's `adj[self.m] = self.e.view(-1)` (:41)."""
    m = (pat > 0).cpu().numpy()
    J = m.shape[0]
    rows, cols = np.nonzero(m)
    nnz = len(rows)
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=J))])
    col_ptr = np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=J))])
    order = np.lexsort((rows, cols))
    # padded fixed-degree (ELL) views for the unrolled kernels: row i -> Dr slots (j, k), column j -> Dc slots (i, k);
    # padding slots point at the row/column itself with edge id nnz, whose weight row is all zero
    Dr = int(np.bincount(rows, minlength=J).max())
    Dc = int(np.bincount(cols, minlength=J).max())
    ell_rj = np.repeat(np.arange(J), Dr).reshape(J, Dr)
    ell_rk = np.full((J, Dr), nnz)
    ell_ci = np.repeat(np.arange(J), Dc).reshape(J, Dc)
    ell_ck = np.full((J, Dc), nnz)
    fill_r = np.zeros(J, dtype=int)
    for k, (i, j) in enumerate(zip(rows, cols)):
        ell_rj[i, fill_r[i]] = j
        ell_rk[i, fill_r[i]] = k
        fill_r[i] += 1
    fill_c = np.zeros(J, dtype=int)
    for q in order:
        i, j = rows[q], cols[q]
        ell_ci[j, fill_c[j]] = i
        ell_ck[j, fill_c[j]] = q
        fill_c[j] += 1
    tab = np.concatenate([[J, nnz], row_ptr, cols, col_ptr, rows[order], order, [Dr, Dc], ell_rj.ravel(), ell_rk.ravel(),
                          ell_ci.ravel(), ell_ck.ravel()]).astype(np.int32)
    return torch.from_numpy(tab), nnz


class SemCHGraphConv(nn.Module):
    """Semantic channel-wise graph convolution layer (parameters only; see module docstring)."""

    def __init__(self, in_features, out_features, adj, bias=False):
        super(SemCHGraphConv, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.W = nn.Parameter(torch.zeros(size=(2, in_features, out_features), dtype=torch.float))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        # plain attributes, not buffers: they must stay out of the state_dict (reference :23-24)
        self.adj = adj.unsqueeze(0).repeat(out_features, 1, 1)
        self.m = (self.adj > 0)
        self.e = nn.Parameter(torch.zeros(out_features, int((adj > 0).sum().item()), dtype=torch.float))
        nn.init.constant_(self.e.data, 1)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features, dtype=torch.float))
            stdv = 1. / math.sqrt(self.W.size(1))
            self.bias.data.uniform_(-stdv, stdv)
        else:
            self.register_parameter('bias', None)

    def forward(self, input):
        """input: (B, T, J, C_in) -> (B, T, J, C_out)   (reference :35-53; forward-only plan, gast_hip/modules.py)"""
        from gast_hip.modules import graph_conv_forward
        return graph_conv_forward(self, input, shared=False)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class LocalGraph(nn.Module):
    def __init__(self, adj, input_dim, output_dim, dropout=None):
        super(LocalGraph, self).__init__()
        adj_sym, adj_con = skeleton_patterns(adj)
        self.gcn_sym = SemCHGraphConv(input_dim, output_dim, adj_sym)
        self.bn_1 = nn.BatchNorm2d(output_dim, momentum=0.1)
        self.gcn_con = SemCHGraphConv(input_dim, output_dim, adj_con)
        self.bn_2 = nn.BatchNorm2d(output_dim, momentum=0.1)
        self.relu = nn.ReLU()
        self.cat_conv = nn.Conv2d(2 * output_dim, output_dim, 1, bias=False)
        self.cat_bn = nn.BatchNorm2d(output_dim, momentum=0.1)
        self.dropout = nn.Dropout(dropout) if dropout is not None else None

    def forward(self, input):
        """input: (B, T, J, C) -> (B, T, J, C_out)   (reference :130-151; forward-only plan, gast_hip/modules.py)"""
        from gast_hip.modules import local_graph_forward
        return local_graph_forward(self, input)
